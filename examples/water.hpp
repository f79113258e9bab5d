// water.hpp -- the UPDATE POLICY of the reference's ocean node (assets/water/water.gd) as a compiled host class on top of
// examples/wave_generator.hpp: SURVEY.md 8f row N1, the step immediately above the hot path.
//
// Mirrored, line for line in behaviour: the `parameters` setter (seed and time offset per cascade, every spectrum dirty, generator
// rebuilt: water.gd:22-35), `map_size` (rebuilds: :38-41), `updates_per_second` (keeps the phase of the next update when the rate
// changes: :51-54), `_process` (rate limiter with catch-up delta: :75-82; then the child WaveGenerator's own _process, one armed
// cascade per rendered frame: wave_generator.gd:56-63), `_setup_wave_generator` (:84-100), `_update_scales_uniform` (:102-110) and
// `_update_water` (:112-114).  NOT mirrored: rendering (mesh, materials, global shader parameters) and Godot's RandomNumberGenerator
// (PCG32, engine source; water.gd:31 draws the seeds from it) -- seeds are explicit inputs.
// Header-only C++17, no Godot types: the body a GDExtension `Water` node wraps.
#pragma once
#include <array>
#include <memory>
#include <optional>
#include <utility>
#include <vector>

#include "wave_generator.hpp"

namespace ocean_waves {

class Water {
public:
    double time = 0.0;              // water.gd:61
    double next_update_time = 0.0;  // water.gd:62
    uint32_t generator_flags = 0;   // OW_FLAG_* handed to every generator this node builds
    int device_id = -1;

    // `parameters = value` (water.gd:22-35).  seeds[i] replaces rng.randi_range(-10000, 10000) x 2 (:31); nullptr keeps the
    // records' own seeds.
    void set_parameters(std::vector<ParametersRef> value, const std::vector<std::pair<int32_t, int32_t>> *seeds = nullptr) {
        for (size_t i = 0; i < value.size(); ++i) {
            if (!value[i]) value[i] = std::make_shared<WaveCascadeParameters>();  // :28 "ensure all values have an associated cascade"
            if (seeds) value[i]->set_spectrum_seed((*seeds)[i].first, (*seeds)[i].second);
            value[i]->set_time(120.0 + 3.14159265358979323846 * (double)i);      // :32 "such that cascades don't interfere"
        }
        parameters_ = std::move(value);
        setup_wave_generator();
    }
    const std::vector<ParametersRef> &parameters() const { return parameters_; }

    int map_size() const { return map_size_; }
    void set_map_size(int value) {  // :38-41
        map_size_ = value;
        setup_wave_generator();
    }

    double updates_per_second() const { return updates_per_second_; }
    void set_updates_per_second(double value) {  // :51-54
        next_update_time = next_update_time - (1.0 / (updates_per_second_ + 1e-10) - 1.0 / (value + 1e-10));
        updates_per_second_ = value;
    }

    // One rendered frame (water.gd:75-82, then the child node's _process).  Returns the delta of the update that was issued.
    std::optional<double> _process(double delta) {
        std::optional<double> issued;
        if (updates_per_second_ == 0.0 || time >= next_update_time) {
            const double target_update_delta = 1.0 / (updates_per_second_ + 1e-10);
            const double update_delta = updates_per_second_ == 0.0 ? delta : target_update_delta + (time - next_update_time);
            next_update_time = time + target_update_delta;
            update_water(update_delta);
            issued = update_delta;
        }
        time += delta;
        if (wave_generator_) wave_generator_->_process(delta);  // child node: its _process runs after the parent's
        return issued;
    }

    // `map_scales` of the water / spray materials (:102-110): (1 / tile_length.x, 1 / tile_length.y, displacement_scale, normal_scale)
    std::vector<std::array<float, 4>> map_scales() const {
        std::vector<std::array<float, 4>> s(parameters_.size());
        for (size_t i = 0; i < parameters_.size(); ++i) {
            const auto tile = parameters_[i]->tile_length();
            s[i] = {1.0f / tile.first, 1.0f / tile.second, (float)parameters_[i]->displacement_scale(), (float)parameters_[i]->normal_scale()};  // Vector4: FP32
        }
        return s;
    }

    WaveGenerator *wave_generator() const { return wave_generator_.get(); }
    int generators_built() const { return generators_built_; }

private:
    void setup_wave_generator() {  // :84-100
        if (parameters_.empty()) return;
        for (auto &p : parameters_) p->set_should_generate_spectrum(true);
        wave_generator_ = std::make_unique<WaveGenerator>();  // (`wave_generator = value` queue_free()s the old node, :56-59)
        wave_generator_->map_size = map_size_;
        wave_generator_->flags = generator_flags;
        wave_generator_->device_id = device_id;
        wave_generator_->init_gpu(parameters_.size() < 2 ? 2 : (int)parameters_.size());  // maxi(2, n), :91
        ++generators_built_;
    }
    void update_water(double delta) {  // :112-114
        if (!wave_generator_) setup_wave_generator();
        if (!wave_generator_) return;  // no parameters yet (the reference would dereference null here)
        wave_generator_->update(delta, parameters_);
    }

    std::vector<ParametersRef> parameters_;
    std::unique_ptr<WaveGenerator> wave_generator_;
    int map_size_ = 1024;                // :38
    double updates_per_second_ = 50.0;   // :51
    int generators_built_ = 0;
};

}  // namespace ocean_waves
