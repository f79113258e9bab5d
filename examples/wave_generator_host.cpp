// wave_generator_host.cpp -- drives examples/wave_generator.hpp the way water.gd drives the reference's WaveGenerator
// (water.gd:89-91,112-114; wave_generator.gd:56-63): update() when nothing is armed, _process() once per "rendered frame",
// finished layers delivered to a texture sink one frame later.  Prints the same line as examples/c_consumer.c so that
// tests/test_cpp_host.py can hold it to the Python mirror.
//   g++ -O2 -std=c++17 -Iinclude examples/wave_generator_host.cpp -o wave_generator_host -Lgodotoceanwaves_amd -locean_waves
//       -Wl,-rpath,$PWD/godotoceanwaves_amd -Wl,-rpath-link,/opt/rocm/lib && ./wave_generator_host [map_size [frames]]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "wave_generator.hpp"

using namespace ocean_waves;

static uint64_t fnv1a(const void *data, size_t n) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char **argv) try {
    const int n = argc > 1 ? atoi(argv[1]) : 256, cascades = 3, frames = argc > 2 ? atoi(argv[2]) : 12;
    // argv[3]: every layer the sink is handed is also written to <prefix><layer>.bin (displacement bytes, then normal bytes; a later
    // hand-off of the same layer replaces the file): the bytes texture_update would get, for the tests to hold to the fixtures
    const std::string dump = argc > 3 ? argv[3] : "";
    // the three cascades of the reference's main.tscn, seeds and time offsets as water.gd:31-32 assigns them
    const float tile[3] = {88.0f, 57.0f, 16.0f}, wind[3] = {10.0f, 5.0f, 20.0f}, dir[3] = {20.0f, 15.0f, 20.0f};
    const float fetch[3] = {150.0f, 150.0f, 550.0f}, spread[3] = {0.2f, 0.4f, 0.4f}, whitecap[3] = {0.5f, 0.5f, 0.25f}, foam[3] = {8.0f, 0.0f, 3.0f};
    std::vector<ParametersRef> parameters;
    float map_scales[3][4];
    for (int i = 0; i < cascades; ++i) {
        auto p = std::make_shared<WaveCascadeParameters>();
        p->set_tile_length(tile[i], tile[i]);
        p->set_wind_speed(wind[i]);
        p->set_wind_direction(dir[i]);
        p->set_fetch_length(fetch[i]);
        p->set_spread(spread[i]);
        p->set_whitecap(whitecap[i]);
        p->set_foam_amount(foam[i]);
        p->set_spectrum_seed(1000 + 17 * i, -2000 + 31 * i);
        p->set_time(120.0 + 3.14159265358979323846 * i);
        parameters.push_back(p);
        map_scales[i][0] = map_scales[i][1] = 1.0f / tile[i];  // water.gd:105-109
        map_scales[i][2] = (float)p->displacement_scale();
        map_scales[i][3] = (float)p->normal_scale();
    }

    WaveGenerator wave_generator;
    wave_generator.map_size = n;                       // water.gd:90
    wave_generator.init_gpu(cascades < 2 ? 2 : cascades);  // water.gd:91

    // the engine side: Texture2DArrayRD pair updated with texture_update (water.gd:93-100); here the sink folds what it is
    // handed into the checksum examples/c_consumer.c computes (displacement first, then normal, of the same layer)
    uint64_t sum = 0, pending = 0;
    int handed = 0;
    wave_generator.set_texture_update([&](const char *which, int layer, const void *bytes, size_t size) {
        if (!dump.empty()) {
            FILE *f = std::fopen((dump + std::to_string(layer) + ".bin").c_str(), std::strcmp(which, "displacement_map") == 0 ? "wb" : "ab");
            if (!f || std::fwrite(bytes, 1, size, f) != size || std::fclose(f) != 0) throw Error(OW_ERR_INVALID, "cannot write the dump");
        }
        if (std::strcmp(which, "displacement_map") == 0) {
            pending = fnv1a(bytes, size);
        } else {
            sum ^= pending + 31 * fnv1a(bytes, size) + (uint64_t)layer;
            ++handed;
        }
    });

    for (int f = 0; f < frames; ++f) {
        if (wave_generator.pass_num_cascades_remaining() == 0) wave_generator.update(1.0 / 50.0, parameters);  // water.gd:114
        wave_generator._process(1.0 / 60.0);           // engine-driven, once per rendered frame
    }
    wave_generator.flush();
    for (auto &p : parameters)
        if (p->should_generate_spectrum()) throw Error(OW_ERR_STATE, "a processed cascade is still marked dirty");

    float xz[64][2];
    ow_surface_sample s[64];
    for (int i = 0; i < 64; ++i) {
        xz[i][0] = -40.0f + 1.25f * i;
        xz[i][1] = 7.5f + 0.5f * i;
    }
    check(ow_sample_surface(wave_generator.context(), &xz[0][0], 64, &map_scales[0][0], cascades, s));
    double hmin = 1e9, hmax = -1e9;
    int active = 0;
    for (int i = 0; i < 64; ++i) {
        hmin = std::fmin(hmin, s[i].displacement[1]);
        hmax = std::fmax(hmax, s[i].displacement[1]);
        active += s[i].spray_active;
    }
    std::printf("layers_handed_off=%d checksum=%016llx wave_height=[%.4f,%.4f] spray_active=%d\n", handed, (unsigned long long)sum, hmin, hmax, active);
    if (!(hmax > hmin) || !std::isfinite(hmin) || !std::isfinite(hmax)) {
        std::fprintf(stderr, "flat or non-finite surface\n");
        return 1;
    }
    return 0;
} catch (const ocean_waves::Error &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
}
