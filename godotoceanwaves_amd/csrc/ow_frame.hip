// ow_frame.hip -- the two per-frame kernels of the ocean-wave hot path for gfx950 (MI355X).
//
//   k_pass1 : h0 + omega --(time modulate, spectrum_modulate.glsl)--> 4 packed spectra
//             --(row IFFT, fft_compute.glsl 1st dispatch)--> transposed store (transpose.glsl fused)
//   k_pass2 : row IFFT (fft_compute.glsl 2nd dispatch) --> fft_unpack.glsl fused (sign, Jacobian,
//             foam RMW, RGBA16F stores)
//
// One workgroup == one wavefront (64 lanes) == plan_rows_per_wave(N) map rows x 4 layers.  There is
// no s_barrier: the two LDS exchanges of a row transform stay inside the wave, whose DS instructions
// execute in order; wave_sync() only pins the compiler's ordering.
#include "ow_kernels.h"

namespace ow {

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// row IFFT of the P points in d[] (lane t of the row), exchanging through this row's LDS buffer
template <int N>
__device__ __forceinline__ void row_ifft(cplx *d, int t, cplx *lds_row, const cplx *__restrict__ tw) {
    fft_stage_compute<N, 0>(d, t, tw);
    fft_stage_write<N, 0>(d, t, lds_row);
    wave_sync();
    fft_stage_read<N, 1>(d, t, lds_row);
    wave_sync();
    fft_stage_compute<N, 1>(d, t, tw);
    if constexpr (plan_S(N) == 3) {
        fft_stage_write<N, 1>(d, t, lds_row);
        wave_sync();
        fft_stage_read<N, 2>(d, t, lds_row);
        wave_sync();
        fft_stage_compute<N, 2>(d, t, tw);
    }
}

// blockIdx -> (launch slot, row group).  Consecutive row groups are kept on one XCD (blocks are
// dealt round-robin to the 8 XCDs) so that the four 32-byte granules of a 128-byte line of T,
// written by four consecutive rows, meet in the same L2.
template <int N>
__device__ __forceinline__ void block_to_rows(int &slot, int &row0) {
    constexpr int RW = plan_rows_per_wave(N), BPC = N / RW;
    const int b = blockIdx.x;
    slot = b / BPC;
    const int r = b % BPC;
    const int rg = (r % 8) * (BPC / 8) + r / 8;
    row0 = rg * RW;
}

template <int N>
__global__ __launch_bounds__(64) void k_pass1(DeviceBuffers buf, FrameArgs args) {
    constexpr int Tn = plan_T(N), P = plan_P(N);
    __shared__ cplx lds[plan_lds_cplx(N)];
    int slot, row0;
    block_to_rows<N>(slot, row0);
    const CascadeFrame cf = args.c[slot];
    const int lane = threadIdx.x, rw = lane / Tn, t = lane % Tn;
    const int y = row0 + rw;
    const size_t plane = (size_t)N * N;
    cplx *lds_row = lds + rw * plan_row_slots(N);

    cplx h[P];
    Pass1<N>::load_modulate(h, t, buf.h0 + cf.cascade * plane + (size_t)y * N,
                            buf.omega + cf.cascade * plane + (size_t)y * N, cf.time);
    const float dkx = (2.0f * kPi) / cf.tile_x, dky = (2.0f * kPi) / cf.tile_y;
    const float ky = (float)(y - N / 2) * dky;

    cplx out[kLayers][P];
    Pass1<N>::template layer_input<0>(out[0], h, t, ky, dkx);
    row_ifft<N>(out[0], t, lds_row, buf.tw);
    Pass1<N>::template layer_input<1>(out[1], h, t, ky, dkx);
    row_ifft<N>(out[1], t, lds_row, buf.tw);
    Pass1<N>::template layer_input<2>(out[2], h, t, ky, dkx);
    row_ifft<N>(out[2], t, lds_row, buf.tw);
    Pass1<N>::template layer_input<3>(out[3], h, t, ky, dkx);
    row_ifft<N>(out[3], t, lds_row, buf.tw);

    Pass1<N>::store(out, t, y, buf.T + cf.cascade * plane * kLayers);
}

template <int N, bool F32>
__global__ __launch_bounds__(64) void k_pass2(DeviceBuffers buf, FrameArgs args) {
    constexpr int Tn = plan_T(N), P = plan_P(N);
    __shared__ cplx lds[plan_lds_cplx(N)];
    int slot, row0;
    block_to_rows<N>(slot, row0);
    const CascadeFrame cf = args.c[slot];
    const int lane = threadIdx.x, rw = lane / Tn, t = lane % Tn;
    const int xp = row0 + rw;
    const size_t plane = (size_t)N * N;
    cplx *lds_row = lds + rw * plan_row_slots(N);

    cplx d[kLayers][P];
    Pass2<N>::load(d, t, xp, buf.T + cf.cascade * plane * kLayers);
    row_ifft<N>(d[0], t, lds_row, buf.tw);
    row_ifft<N>(d[1], t, lds_row, buf.tw);
    row_ifft<N>(d[2], t, lds_row, buf.tw);
    row_ifft<N>(d[3], t, lds_row, buf.tw);

    const size_t row_off = cf.cascade * plane + (size_t)xp * N;
    Pass2<N>::unpack_store(d, t, xp, cf, buf.disp + row_off, buf.norm + row_off,
                           F32 ? buf.f32 + row_off * 8 : nullptr);
}

bool supported_map_size(int n) { return n == 128 || n == 256 || n == 512 || n == 1024 || n == 2048; }

template <int N>
static hipError_t launch1(int slots, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s) {
    const int blocks = slots * (N / plan_rows_per_wave(N));
    hipLaunchKernelGGL((k_pass1<N>), dim3(blocks), dim3(64), 0, s, buf, args);
    return hipGetLastError();
}
template <int N>
static hipError_t launch2(int slots, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s) {
    const int blocks = slots * (N / plan_rows_per_wave(N));
    if (buf.f32) hipLaunchKernelGGL((k_pass2<N, true>), dim3(blocks), dim3(64), 0, s, buf, args);
    else hipLaunchKernelGGL((k_pass2<N, false>), dim3(blocks), dim3(64), 0, s, buf, args);
    return hipGetLastError();
}

hipError_t launch_pass1(int n, int slots, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s) {
    switch (n) {
        case 128: return launch1<128>(slots, args, buf, s);
        case 256: return launch1<256>(slots, args, buf, s);
        case 512: return launch1<512>(slots, args, buf, s);
        case 1024: return launch1<1024>(slots, args, buf, s);
        case 2048: return launch1<2048>(slots, args, buf, s);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_pass2(int n, int slots, const FrameArgs &args, const DeviceBuffers &buf, hipStream_t s) {
    switch (n) {
        case 128: return launch2<128>(slots, args, buf, s);
        case 256: return launch2<256>(slots, args, buf, s);
        case 512: return launch2<512>(slots, args, buf, s);
        case 1024: return launch2<1024>(slots, args, buf, s);
        case 2048: return launch2<2048>(slots, args, buf, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace ow
