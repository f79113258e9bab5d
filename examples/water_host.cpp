// water_host.cpp -- drives examples/water.hpp (the ocean node's update policy, water.gd) from a small frame script, so that a test
// can hold the compiled scheduler + the real kernels to the oracle advanced on the same schedule (tests/test_water_host.py).
//   g++ -O2 -std=c++17 -Iinclude -Iexamples examples/water_host.cpp -o water_host -Lgodotoceanwaves_amd -locean_waves
//       -Wl,-rpath,$PWD/godotoceanwaves_amd -Wl,-rpath-link,/opt/rocm/lib && ./water_host script.txt
// Script, one command per line:
//   mapsize N | params C | frame DELTA | rate UPS | wind I V | foam I V | tile I X Y | dump FILE
// `params C` = the first C rows of SURVEY.md 8d's cascade table (0-2: the reference's main.tscn) with its explicit seeds.
// Output: one "update <delta> time <node time>" line per issued update (%.17g), "generators <count>", and `dump` writes
// [C][2 maps][N][N][4] uint16 after draining what is armed the way further rendered frames would.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

#include "water.hpp"

using namespace ocean_waves;

static const float kTable[8][9] = {  // tile, U, dir, fetch km, swell, spread, detail, whitecap, foam  
    {88, 10, 20, 150, 0.8f, 0.2f, 1, 0.5f, 8},  {57, 5, 15, 150, 0.8f, 0.4f, 1, 0.5f, 0},  {16, 20, 20, 550, 0.8f, 0.4f, 1, 0.25f, 3},
    {250, 20, 0, 550, 0.8f, 0.2f, 1, 0.5f, 5},  {33, 20, 0, 550, 0.8f, 0.2f, 1, 0.5f, 5},  {137, 20, 0, 550, 0.8f, 0.2f, 1, 0.5f, 5},
    {23, 20, 0, 550, 0.8f, 0.2f, 1, 0.5f, 5},   {9, 20, 0, 550, 0.8f, 0.2f, 1, 0.5f, 5}};

int main(int argc, char **argv) try {
    if (argc < 2) {
        std::fprintf(stderr, "usage: water_host script.txt\n");
        return 2;
    }
    std::ifstream in(argv[1]);
    if (!in) {
        std::fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    Water water;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string cmd;
        if (!(ls >> cmd) || cmd[0] == '#') continue;
        if (cmd == "mapsize") {
            int n;
            ls >> n;
            water.set_map_size(n);
        } else if (cmd == "params") {
            int c;
            ls >> c;
            std::vector<ParametersRef> ps;
            std::vector<std::pair<int32_t, int32_t>> seeds;
            for (int i = 0; i < c; ++i) {
                auto p = std::make_shared<WaveCascadeParameters>();
                const float *r = kTable[i];
                p->set_tile_length(r[0], r[0]);
                p->set_wind_speed(r[1]);
                p->set_wind_direction(r[2]);
                p->set_fetch_length(r[3]);
                p->set_swell(r[4]);
                p->set_spread(r[5]);
                p->set_detail(r[6]);
                p->set_whitecap(r[7]);
                p->set_foam_amount(r[8]);
                ps.push_back(p);
                seeds.emplace_back(1000 + 17 * i, -2000 + 31 * i);
            }
            water.set_parameters(ps, &seeds);
        } else if (cmd == "frame") {
            double d;
            ls >> d;
            if (auto issued = water._process(d)) std::printf("update %.17g time %.17g\n", *issued, water.time);
        } else if (cmd == "rate") {
            double r;
            ls >> r;
            water.set_updates_per_second(r);
        } else if (cmd == "wind" || cmd == "foam") {
            int i;
            double v;  // (a GDScript float is FP64: the edit reaches the library un-narrowed)
            ls >> i >> v;
            if (cmd == "wind") water.parameters().at(i)->set_wind_speed(v);  // a live edit: the setter raises the dirty flag
            else water.parameters().at(i)->set_foam_amount(v);
        } else if (cmd == "tile") {
            int i;
            float x, y;
            ls >> i >> x >> y;
            water.parameters().at(i)->set_tile_length(x, y);
        } else if (cmd == "dump") {
            std::string path;
            ls >> path;
            WaveGenerator *g = water.wave_generator();
            if (!g) throw Error(OW_ERR_STATE, "dump before any parameters");
            g->sync();
            const size_t n = (size_t)water.map_size(), bytes = n * n * 8;
            std::vector<char> d(bytes), m(bytes);
            FILE *f = std::fopen(path.c_str(), "wb");
            if (!f) throw Error(OW_ERR_INVALID, "cannot write " + path);
            for (size_t c = 0; c < water.parameters().size(); ++c) {
                check(ow_get_maps(g->context(), (int32_t)c, d.data(), m.data()));
                std::fwrite(d.data(), 1, bytes, f);
                std::fwrite(m.data(), 1, bytes, f);
            }
            std::fclose(f);
            for (auto &p : water.parameters()) std::printf("cascade_time %.17g dirty %d\n", p->time(), p->should_generate_spectrum() ? 1 : 0);
            const auto scales = water.map_scales();
            for (auto &s : scales) std::printf("map_scale %.9g %.9g %.9g %.9g\n", s[0], s[1], s[2], s[3]);
            std::printf("remaining %d generators %d next_update_time %.17g\n", g->pass_num_cascades_remaining(), water.generators_built(), water.next_update_time);
        } else {
            std::fprintf(stderr, "unknown command: %s\n", cmd.c_str());
            return 2;
        }
    }
    return 0;
} catch (const ocean_waves::Error &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
}
