// wave_generator.hpp -- a compiled host class in the shape of the reference's WaveGenerator / WaveCascadeParameters
// (assets/water/wave_generator.gd, assets/water/wave_cascade_parameters.gd) on top of the C-ABI (include/ocean_waves.h).
//
// This is the body a GDExtension node would wrap: same members (map_size, init_gpu, update, _process, descriptors,
// the JONSWAP statics), RAII over ow_context, parameter objects that are LIVE the way Godot Resources are (the generator
// keeps shared references, wave_generator.gd:108, and reads the objects when it processes a cascade, :56-72), and the
// hand-off of finished RGBA16F layers to the engine's textures (water.gd:93-100: Texture2DArrayRD bound as global shader
// uniforms; here a sink callback that receives the bytes RenderingDevice.texture_update(tex, layer, bytes) takes), pipelined
// one frame behind the compute exactly as INTEGRATION.md section 2 lays out.
// Header-only C++17; needs nothing but the C header.  No Godot types: a GDExtension adds the ClassDB bindings around it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ocean_waves.h"

namespace ocean_waves {

class Error : public std::runtime_error {
public:
    Error(ow_status st, const std::string &what) : std::runtime_error(what), status(st) {}
    ow_status status;
};
inline void check(ow_status st) {
    if (st != OW_OK) throw Error(st, std::string("ocean_waves: ") + ow_last_error());
}

// wave_cascade_parameters.gd:7-42.  The exported properties are private fields behind setters that raise
// should_generate_spectrum, as the GDScript `set(value)` blocks do; wind_speed / fetch_length clamp at 1e-4 (:15,:20).
// GDScript floats are FP64 and so are these (ow_cascade_params, ABI 4): the library narrows where the reference does.
class WaveCascadeParameters {
public:
    WaveCascadeParameters() { ow_cascade_params_default(&c_); }

#define OW_EXPORT(name, expr)                  \
    double name() const { return c_.name; }    \
    void set_##name(double value) {            \
        c_.name = (expr);                      \
        c_.should_generate_spectrum = 1;       \
    }
    OW_EXPORT(wind_speed, value < 1e-4 ? 1e-4 : value)       // :15
    OW_EXPORT(wind_direction, value)                           // :17
    OW_EXPORT(fetch_length, value < 1e-4 ? 1e-4 : value)     // :20
    OW_EXPORT(swell, value)                                    // :22
    OW_EXPORT(spread, value)                                   // :25
    OW_EXPORT(detail, value)                                   // :28
    OW_EXPORT(whitecap, value)                                 // :32
    OW_EXPORT(foam_amount, value)                              // :34
#undef OW_EXPORT
    std::pair<float, float> tile_length() const { return {c_.tile_length[0], c_.tile_length[1]}; }
    void set_tile_length(float x, float y) {                   // :7
        c_.tile_length[0] = x;
        c_.tile_length[1] = y;
        c_.should_generate_spectrum = 1;
    }
    // consumer-side only: no dirty flag (:9-12)
    double displacement_scale() const { return c_.displacement_scale; }
    void set_displacement_scale(double v) { c_.displacement_scale = v; }
    double normal_scale() const { return c_.normal_scale; }
    void set_normal_scale(double v) { c_.normal_scale = v; }

    // plain vars of the resource (:37-42)
    std::pair<int32_t, int32_t> spectrum_seed() const { return {c_.spectrum_seed[0], c_.spectrum_seed[1]}; }
    void set_spectrum_seed(int32_t x, int32_t y) {
        c_.spectrum_seed[0] = x;
        c_.spectrum_seed[1] = y;
    }
    bool should_generate_spectrum() const { return c_.should_generate_spectrum != 0; }
    void set_should_generate_spectrum(bool v) { c_.should_generate_spectrum = v ? 1 : 0; }
    double time() const { return c_.time; }
    void set_time(double t) { c_.time = t; }
    double foam_grow_rate() const { return c_.foam_grow_rate; }
    double foam_decay_rate() const { return c_.foam_decay_rate; }

    const ow_cascade_params &record() const { return c_; }

private:
    friend class WaveGenerator;
    ow_cascade_params c_;
};
using ParametersRef = std::shared_ptr<WaveCascadeParameters>;  // Godot Resources are reference counted

class WaveGenerator {
public:
    // RenderingContext.Descriptor stand-in (render_context.gd:23-28): `.rid` is the device pointer of the array texture
    struct Descriptor {
        void *rid = nullptr;
        size_t layer_stride = 0;  // bytes between layers (= map_size^2 * 8)
    };
    // receives one finished layer: which = "displacement_map" / "normal_map" (the keys of `descriptors`, wave_generator.gd:34-35),
    // bytes = map_size^2 * 8 of RGBA16F in RenderingDevice.texture_update(texture, layer, bytes) layout; valid during the call
    using TextureUpdate = std::function<void(const char *which, int layer, const void *bytes, size_t size)>;

    int map_size = 0;          // wave_generator.gd:8
    float depth = 20.0f;       // DEPTH, :6
    uint32_t flags = 0;        // OW_FLAG_*
    int device_id = -1;

    WaveGenerator() = default;
    WaveGenerator(const WaveGenerator &) = delete;
    WaveGenerator &operator=(const WaveGenerator &) = delete;
    ~WaveGenerator() { free(); }  // NOTIFICATION_PREDELETE -> context.free(), :111-113

    // :17-54
    void init_gpu(int num_cascades) {
        free();
        ow_config cfg{};
        cfg.map_size = map_size;
        cfg.num_cascades = num_cascades;
        cfg.device_id = device_id;
        cfg.depth = depth;
        cfg.flags = flags;
        check(ow_create(&cfg, &context_));
        num_cascades_ = num_cascades;
        size_t stride = 0;
        check(ow_get_device_ptrs(context_, &displacement_map_.rid, &normal_map_.rid, &stride));
        displacement_map_.layer_stride = normal_map_.layer_stride = stride;
    }

    // descriptors[&'displacement_map'], descriptors[&'normal_map'] (:11,34-35; read by water.gd:95-96)
    const Descriptor &displacement_map() const { return displacement_map_; }
    const Descriptor &normal_map() const { return normal_map_; }
    int pass_num_cascades_remaining() const { return context_ ? ow_cascades_remaining(context_) : 0; }  // :15

    // Where finished layers go when the consumer is not on this device: the engine-side texture_update (water.gd:93-100
    // creates the Texture2DArrayRD pair; the maps are CAN_UPDATE, render_context.gd:76-85).  Without a sink nothing leaves
    // the device (zero-copy consumers read the descriptors).
    void set_texture_update(TextureUpdate sink) { sink_ = std::move(sink); }

    // :90-109
    void update(double delta, const std::vector<ParametersRef> &parameters) {
        if (parameters.empty()) throw Error(OW_ERR_INVALID, "update(): parameters must not be empty");  // assert, :91
        if (!context_) init_gpu(parameters.size() < 2 ? 2 : (int)parameters.size());                   // :92-93
        const int leftovers = pass_num_cascades_remaining();
        for (int i = 0; i < leftovers; ++i) push_live(i);  // the flush (:94-98) reads the live objects
        std::vector<ow_cascade_params> records(parameters.size());
        for (size_t i = 0; i < parameters.size(); ++i) {
            records[i] = parameters[i]->c_;
            for (int j = 0; j < leftovers; ++j)  // flushed by this very call: its spectrum is regenerated there, not twice
                if (pass_parameters_[j] == parameters[i]) records[i].should_generate_spectrum = 0;
        }
        check(ow_update(context_, delta, records.data(), (int32_t)records.size()));
        for (int j = 0; j < leftovers; ++j) pass_parameters_[j]->c_.should_generate_spectrum = 0;  // :72, via the flush
        for (size_t i = 0; i < parameters.size(); ++i) {  // :103-106: the generator advances these inside the resource
            parameters[i]->c_.time = records[i].time;
            parameters[i]->c_.foam_grow_rate = records[i].foam_grow_rate;
            parameters[i]->c_.foam_decay_rate = records[i].foam_decay_rate;
        }
        pass_parameters_ = parameters;  // :108 (shared references: the objects stay live)
    }

    // :56-63 -- one armed cascade per rendered frame, highest index first; with a sink, the layer computed by the PREVIOUS
    // call is delivered first (it crossed PCIe while that frame rendered), then this frame's layer starts its way out
    void _process(double /*delta*/) {
        if (!context_) return;
        deliver();
        const int remaining = pass_num_cascades_remaining();
        if (remaining == 0) return;
        const int layer = remaining - 1;
        push_live(layer);
        check(ow_process(context_));
        pass_parameters_[layer]->c_.should_generate_spectrum = 0;  // :72
        if (sink_) {
            check(ow_readback_begin(context_, 1u << layer));  // device-side snapshot + asynchronous copy: does not block
            in_flight_ = layer;
        }
    }

    // delivers a layer still on its way (call before reading the sink's textures outside the frame loop)
    void flush() { deliver(); }
    void sync() { check(ow_sync(context_)); }
    ow_context *context() const { return context_; }  // :9

    // :111-113
    void free() {
        if (context_) ow_destroy(context_);
        context_ = nullptr;
        in_flight_ = -1;
        pass_parameters_.clear();
    }

    // :116-121
    static double JONSWAP_alpha(double wind_speed = 20.0, double fetch_length = 550e3) { return ow_jonswap_alpha(wind_speed, fetch_length); }
    static double JONSWAP_peak_angular_frequency(double wind_speed = 20.0, double fetch_length = 550e3) {
        return ow_jonswap_peak_angular_frequency(wind_speed, fetch_length);
    }

private:
    void push_live(int index) { check(ow_set_cascade_params(context_, index, &pass_parameters_[index]->c_)); }
    void deliver() {
        if (in_flight_ < 0) return;
        const void *d = nullptr, *m = nullptr;
        check(ow_readback_wait(context_, in_flight_, &d, &m));
        const size_t bytes = (size_t)map_size * map_size * 8;
        if (sink_) {
            sink_("displacement_map", in_flight_, d, bytes);
            sink_("normal_map", in_flight_, m, bytes);
        }
        in_flight_ = -1;
    }

    ow_context *context_ = nullptr;
    int num_cascades_ = 0, in_flight_ = -1;
    Descriptor displacement_map_, normal_map_;
    std::vector<ParametersRef> pass_parameters_;  // :14
    TextureUpdate sink_;
};

}  // namespace ocean_waves
