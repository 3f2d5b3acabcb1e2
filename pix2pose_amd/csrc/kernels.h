// Internal launch interface between the host-side model code and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace p2p {

// Route switches between kernels that compute the same bits (P2P_NO_HALO, P2P_STREAM_WGS, P2P_NO_FUSED_BLOCK, ...) exist in DEVELOPMENT builds
// only: -DP2P_DEV_SWITCHES = pix2pose_amd/libp2p_mi355_dev.so, which the route-equivalence tests and the A/B tools load through P2P_LIB.
// The shipped library takes no behaviour from the environment (a stray variable cannot silently change its speed).
#ifdef P2P_DEV_SWITCHES
inline const char* dev_env(const char* name) { return getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

// Measurement hook of the glue / PnP launches (p2p_profile_*, slots 12..19 of p2p_mi355.h): the entry points of the pipeline set it while
// profiling is enabled on their context; a ProfScope around a launch then brackets it with HIP events on the stream it is launched on.
struct ProfHook {
    void* ctx;
    void (*begin)(void* ctx, int slot, hipStream_t s);
    void (*end)(void* ctx, hipStream_t s);
};
extern thread_local ProfHook g_prof_hook;
struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(int slot, hipStream_t st) : s(st), on(g_prof_hook.ctx != nullptr) { if (on) g_prof_hook.begin(g_prof_hook.ctx, slot, s); }
    ~ProfScope() { if (on) g_prof_hook.end(g_prof_hook.ctx, s); }
};

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };
enum EpiMode { EPI_NORMAL = 0, EPI_HEAD = 1 };
enum Prec { PREC_F32 = 0, PREC_F16X3 = 1 };   // igemm arithmetic: fp32 MFMA, or fp32 emulated with 3 split-f16 MFMAs

// Operand-range guard of the split-f16 arithmetic (PREC_F16X3): an activation is split as hi = f16(x), lo = f16(x - hi), so |x| must stay
// below the f16 maximum (65504).  Every epilogue that writes a tensor a later layer reads as a matrix operand tracks the largest magnitude
// it stores (two VALU instructions per float4) and, only if that exceeds the limit, raises a device-side word with atomicMax (float bits of
// a non-negative value order like unsigned integers).  No atomic is issued on the normal path.  Batch-normalised networks sit orders of
// magnitude below; the two linear Dense layers (ae_model.py:199-200) have no BatchNorm behind them.
constexpr float RANGE_LIMIT = 6.0e4f;
// A NaN counts as out of range too (NaN weights or inputs, inf - inf in a residual add): the running maximum is taken with the IEEE-754-2019
// `maximum` (v_maximum3_f32 on gfx950: NaN-propagating, one instruction for two new values), the epilogues' ReLU likewise (relu_nan: fmaxf
// would turn a NaN into 0 and hide it from every later layer), and range_commit raises a NaN as +inf.
// -DP2P_NO_RANGE_GUARD (A/B builds only, tools/ab_round.sh): the guard compiled out, to MEASURE what it costs.
__device__ __forceinline__ float relu_nan(float v) { return __builtin_elementwise_maximum(v, 0.f); }
#ifdef P2P_NO_RANGE_GUARD
template <typename V4> __device__ __forceinline__ float range_note4(float amax, const V4&) { return amax; }
__device__ __forceinline__ float range_note1(float amax, float) { return amax; }
__device__ __forceinline__ void range_commit(unsigned*, float) {}
#else
__device__ __forceinline__ float range_note1(float amax, float v) { return __builtin_elementwise_maximum(amax, fabsf(v)); }
template <typename V4> __device__ __forceinline__ float range_note4(float amax, const V4& v)
{
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_elementwise_maximum(amax, fabsf(v[0])), fabsf(v[1])),
                                         __builtin_elementwise_maximum(fabsf(v[2]), fabsf(v[3])));
}
__device__ __forceinline__ void range_commit(unsigned* acc, float amax)
{
    if (acc && !(amax <= RANGE_LIMIT)) atomicMax(acc, amax == amax ? __float_as_uint(amax) : 0x7F800000u);
}
#endif

constexpr int IGEMM_MAX_TAPS = 25;
constexpr int IGEMM_BK = 32;
constexpr int IGEMM_MAX_GROUPS = 48;  // object models one launch can serve (mixed-object batches)          // K-step (floats); every channel segment is a multiple of it

// One channel segment of the (virtually concatenated) NHWC input: channels
// [coff, coff+C) of a tensor whose pixel stride is `cstride` floats.
struct IgemmSeg {
    const float* ptr;
    int C;
    int cstride;
    int coff;
};

// One object's weight panel inside a grouped launch: GEMM rows [row0, next.row0) use it; its first
// M-tile is tile0 (every group starts on a tile boundary, the last tile of a group may be partial).
struct IgemmGroup {
    const float* w;
    const float* scale;
    const float* shift;
    int row0;
    int tile0;
};

// Implicit-GEMM convolution  D[m, co] = sum_{tap, ci} X[pix(m) + tap][ci] * W[co][tap*Cin + ci]
//   m enumerates (n, gy, gx) over an Hg x Wg grid; input pixel = (gy*in_stride + dy, gx*in_stride + dx)
//   output pixel = (gy*os + oy, gx*os + ox) of an Hout x Wout image (os=2 for transposed-conv phases).
struct IgemmParams {
    IgemmSeg seg[2];
    int seg1_Hin, seg1_Win, seg1_stride;   // 1-tap layers only: segment 1 lives on its own grid (pixel (gy, gx) * seg1_stride of an
                           // seg1_Hin x seg1_Win tensor) -- a strided projection shortcut folded in as extra K; 0 = same grid as segment 0
    int seg0_chunks;       // seg[0].C / 32
    int chunks_per_tap;    // (seg[0].C + seg[1].C) / 32
    int N, Hin, Win;
    int Hg, Wg, M;
    int in_stride;
    int ntaps;
    int8_t dy[IGEMM_MAX_TAPS + 3];
    int8_t dx[IGEMM_MAX_TAPS + 3];
    unsigned seg_bytes[2]; // byte extent of each segment's tensor (buffer-descriptor range)
    unsigned w_bytes;      // byte extent of the weight panel
    const float* w;        // [Cout_pad][K], K = ntaps * Cin_total, rows padded to a multiple of 128
    int K;
    int Cout;
    int ksteps;            // K / 32
    int ksplit;            // gridDim.y; >1 => raw partial sums go to `partial`
    float* partial;        // [ksplit][M][Cout]
    const float* scale;    // per-cout (folded BatchNorm), may be null => 1
    const float* shift;    // per-cout (bias / folded BN shift)
    const float* residual; // optional, indexed [out_pixel * res_cstride + co]
    int res_cstride;
    int act;
    float alpha;
    float* out;
    int Hout, Wout, os, oy, ox;
    int out_cstride, out_coff;
    int mode;
    int prec;              // Prec; with PREC_F16X3 `w` (and grp[].w) is the split-f16 panel, scale carries 2^-10
    // grouped launch (n_groups > 1): detections of several objects in one batch, sorted by object;
    // grp[n_groups] is a sentinel {row0 = M, tile0 = number of M-tiles}.  n_groups <= 1: w/scale/shift above.
    int n_groups;
    unsigned* range_acc;   // operand-range guard (see RANGE_LIMIT): raised when a stored magnitude exceeds the limit; may be null
    IgemmGroup grp[IGEMM_MAX_GROUPS + 1];
};

// tile configurations (BM x BN): 0 = 128x128, 1 = 128x64, 2 = 128x32
hipError_t launch_igemm(const IgemmParams& p, int cfg, hipStream_t s);
int igemm_tile_m(int cfg);   // BM of a tile configuration

// Halo-tiled variant for stride-1 multi-tap layers (igemm_halo.hip): the input patch of a channel slice is staged
// once for all taps.  Same parameter block; PREC_F16X3, grid a multiple of 8x16, no split-K.
bool igemm_halo_supported(const IgemmParams& p);
hipError_t launch_igemm_halo(const IgemmParams& p, hipStream_t s);

// Halo-tiled variant for the layers on an 8x8 output grid (igemm_halo8.hip): the stride-2 5x5 convolution conv4 (through parity
// planes) and the phases of the first transposed convolution.  igemm_halo8_mode(): 0 = not for this kernel.
int igemm_halo8_mode(const IgemmParams& p);
hipError_t launch_igemm_halo8(const IgemmParams& p, hipStream_t s);

// Halo-tiled variant for Conv2D 5x5 stride 2 'SAME' on 16x16 and larger output grids (igemm_halo_s2.hip; parity planes like
// igemm_halo8.hip, 8x16 patches like igemm_halo.hip): the "paper" encoder's conv2 / conv3.
bool igemm_halo_s2_supported(const IgemmParams& p);
hipError_t launch_igemm_halo_s2(const IgemmParams& p, hipStream_t s);

// One ResNet identity bottleneck block (1x1 -> 3x3 -> 1x1 + residual, resnet50_mod.py:40-73) as ONE kernel (resblock.hip): both
// intermediates stay in LDS, the block input is read (with its 3x3 halo) and the output written once.  PREC_F16X3; the same bits as the
// three launches it replaces.  Mixed-object batches: samples [grp[g].sample0, grp[g + 1].sample0) use group g's panels.
struct ResBlockGroup {
    const float* w2a;      // split-f16 panels of the block's three layers (pack_conv)
    const float* w2b;
    const float* w2c;
    const float* ss;       // folded BatchNorm of the three layers: [scale 2a F1 | shift 2a F1 | scale 2b F1 | shift 2b F1 | scale 2c C | shift 2c C]
    int sample0;
    int pad_;
    const float* w2b_frag; // the 2b panel once more in MFMA fragment order [F1 / 32][9 F1 / 32 K-steps (tap, slice)][k half 2][hi, lo][lane 64][8 halves]:
                           // phase B reads its weight operand straight from global (L2) into registers, no LDS staging, no barrier in its K loop
};
struct ResBlockParams {
    const float* x;        // [N, H, W, C] block input (C = 4 F1) -- also the residual
    float* out;            // [N, H, W, C]
    int N, H, W;
    unsigned x_bytes, wa_bytes, wb_bytes, wc_bytes;      // buffer-descriptor ranges
    unsigned* range_acc;   // operand-range guard (RANGE_LIMIT); may be null
    int n_groups;          // >= 1
    ResBlockGroup grp[IGEMM_MAX_GROUPS + 1];
};
bool resblock_supported(int F1, int H, int W);
int resblock_grid(int F1, int N, int H, int W);        // workgroups of the launch
hipError_t launch_resblock(const ResBlockParams& p, int F1, hipStream_t s);
hipError_t launch_resproj(const ResBlockParams& p, int F1, hipStream_t s);      // the projection blocks res2a / res3a (H, W = output grid)

// Winograd F(4,5) along the row axis for Conv2D 5x5 stride 1 'SAME' over a two-segment channel concatenation (wino.hip): 2.5x fewer MFMA
// products than the direct form.  Two launches: the input transform writes V (split-f16, 8 bytes per input element: two positions per input
// column), the GEMM kernel consumes it.  U = the layer's Winograd panel (model.hip: pack_wino), fragment order:
//   [Cout / 64][position 8][Cin / 16][ky 5][(32-channel tile 0 hi, tile 0 lo, tile 1 hi, tile 1 lo)][lane 64][8 halves]
struct WinoGroup {
    const float* U;
    const float* scale;
    const float* shift;
    int sample0;
    int unit0;             // 16x16 grids (two samples per workgroup): first pair of the group -- pairs never straddle two objects
};
struct WinoParams {
    IgemmSeg seg[2];       // input: channels [coff, coff + C) of two NHWC tensors on the same H x W grid
    unsigned seg_bytes[2];
    int seg0_groups;       // seg[0].C / 32
    int N, H, W, Cin, Cout;
    float* V;              // wino_v_bytes(N, H, W, Cin)
    const float* U;
    const float* scale;    // folded BatchNorm (carries the inverse of the panel's per-channel pre-scale)
    const float* shift;
    int act;
    float alpha;
    float* out;            // [N, H, W, out_cstride], channels [out_coff, out_coff + Cout)
    int out_cstride, out_coff;
    unsigned* range_acc;   // operand-range guard: the transform reports the TRANSFORMED operand, the GEMM epilogue what it stores
    int n_groups;          // mixed-object batches: samples [grp[g].sample0, grp[g + 1].sample0) use group g's panel
    WinoGroup grp[IGEMM_MAX_GROUPS + 1];
    int ksplit;            // 1, or the channel slices cut into ksplit ranges (launches under one workgroup per CU): raw sums after the -- linear --
    float* partial;        //   inverse transform go to `partial` [ksplit][N * H * W][Cout], then launch_splitk_reduce (scale / shift / activation)
};
bool wino_supported(int H, int W, int Cin0, int Cin1, int Cout);
size_t wino_v_bytes(int N, int H, int W, int Cin);
int wino_gemm_grid(const WinoParams& p);
hipError_t launch_wino_input(const WinoParams& p, hipStream_t s);
hipError_t launch_wino_gemm(const WinoParams& p, hipStream_t s);

// Winograd F(4,3) along the row axis for Conv2DTranspose 5x5 stride 2 'SAME' (wino3.hip): the four sub-pixel phases as 3-tap correlations on
// the input grid sharing ONE input transform -- 15 position-products per input pixel instead of 25.  Two launches like wino.hip: the input
// transform writes V (split-f16, 6 bytes per input element), the GEMM kernel serves all four phases.  U (model.hip: pack_wino3), fragment order:
//   [py 2][px 2][Cout / 64][position 6][Cin / 16][ky 2 + py][(32-channel half 0 hi, half 0 lo, half 1 hi, half 1 lo)][lane 64][8 halves]
struct Wino3Params {
    const float* in;       // [N, H, W, Cin]
    unsigned in_bytes;
    int N, H, W, Cin, Cout;
    float* V;              // wino3_v_bytes(N, H, W, Cin)
    const float* U;
    const float* scale;    // [4 phases][Cout]: folded BatchNorm times the inverse of the phase panel's per-channel pre-scale
    const float* shift;    // [Cout]
    int act;
    float alpha;
    float* out;            // [N, 2H, 2W, out_cstride], channels [out_coff, out_coff + Cout)
    int out_cstride, out_coff;
    unsigned* range_acc;
    int n_groups;          // mixed-object batches, as in WinoParams (grp[g].scale = that object's [4][Cout] array)
    WinoGroup grp[IGEMM_MAX_GROUPS + 1];
};
bool wino3_supported(int H, int W, int Cin, int Cout);
size_t wino3_v_bytes(int N, int H, int W, int Cin);
hipError_t launch_wino3_input(const Wino3Params& p, hipStream_t s);
hipError_t launch_wino3_gemm(const Wino3Params& p, hipStream_t s);

// The same F(4,3) form for the two stride-2 layers on an 8x8 grid (wino3o.hip): mode 0 = Conv2DTranspose 5x5 / 2 (up1: 8x8 -> 16x16, panel of
// pack_wino3), mode 1 = Conv2D 5x5 / 2 'SAME' (conv4: 16x16 -> 8x8 through the four parity planes, panel of pack_wino3_s2:
//   [Cout / 64][position 6][K-steps: even rows (column parity, Cin / 16, ky 2), odd rows (column parity, Cin / 16, ky 3)][fragment 4][lane 64][8 halves]).
// A workgroup tile is eight samples of one object (grp[g].unit0 = first unit of object g).
struct Wino3oParams {
    const float* in;       // mode 0: [N, 8, 8, Cin]; mode 1: [N, 16, 16, Cin]
    unsigned in_bytes;
    int N, Cin, Cout;
    float* V;              // wino3o_v_bytes(mode, units, Cin)
    const float* U;
    const float* scale;    // mode 0: [4 phases][Cout]; mode 1: [Cout]
    const float* shift;    // [Cout]
    int act;
    float alpha;
    float* out;            // mode 0: [N, 16, 16, out_cstride]; mode 1: [N, 8, 8, out_cstride]
    int out_cstride, out_coff;
    int ksplit;            // mode 1: 1, or 4 = one workgroup per parity plane, raw sums into `partial` [4][N * 64][Cout] (then launch_splitk_reduce)
    float* partial;
    unsigned* range_acc;
    int n_groups;
    WinoGroup grp[IGEMM_MAX_GROUPS + 1];
};
bool wino3o_supported(int mode, int Cin, int Cout);
size_t wino3o_v_bytes(int mode, int units, int Cin);
int wino3o_units(const Wino3oParams& p);
hipError_t launch_wino3o_input(const Wino3oParams& p, int mode, hipStream_t s);
hipError_t launch_wino3o_gemm(const Wino3oParams& p, int mode, hipStream_t s);

// Small-batch variant (igemm_stream.hip): one wave per 32x32 / 64x32 output tile, operands streamed global -> registers with a deep
// software pipeline.  Bit-identical to the batched kernel that serves the layer: every output element is the same chain of MFMAs over
// the same K-step order, which StreamOrder describes as that kernel's loop nest -- for g in groups: for slice: for tap in group g.
//   igemm.hip:                  (tap, slice)          = one group per tap
//   igemm_halo.hip / halo8<0>:  (slice, tap)          = one group holding every tap
//   halo8<1> / igemm_halo_s2:   (plane, slice, tap)   = the four parity planes' tap lists
struct StreamOrder {
    int n_groups;
    int8_t gstart[IGEMM_MAX_TAPS + 3];     // group g = entries [gstart[g], gstart[g + 1]) of tap[]
    int8_t tap[IGEMM_MAX_TAPS + 3];        // panel tap indices
};
// What differs between the sub-problems of a merged launch (the four phases of a transposed convolution).
struct StreamPhase {
    const float* w;
    const float* scale;
    const float* shift;
    unsigned w_bytes;
    int K, ntaps, oy, ox;
    int8_t dy[IGEMM_MAX_TAPS + 3];
    int8_t dx[IGEMM_MAX_TAPS + 3];
    StreamOrder o;
};
struct StreamMulti {
    int n;
    StreamPhase ph[4];
};
bool igemm_stream_supported(const IgemmParams& p);
int igemm_stream_waves(const IgemmParams& p, int tm);      // waves of a streaming launch with 32 tm x 32 tiles (before any K split)
void stream_phase_of(const IgemmParams& p, const StreamOrder& o, StreamPhase* ph);
hipError_t launch_igemm_stream(const IgemmParams& p, const StreamMulti& mp, hipStream_t s);

// Halo-tiled kernel for the merged output heads (heads.hip); takes the same parameter block as the
// generic kernel when heads_halo_supported() says so (PREC_F16X3, 64-wide grid, 128 input channels).
bool heads_halo_supported(const IgemmParams& p);
hipError_t launch_heads_halo(const IgemmParams& p, hipStream_t s);

// out[m][co] = act(sum_z partial[z][m][co] * scale + shift)
hipError_t launch_splitk_reduce(const float* partial, int ksplit, int M, int Cout, const float* scale,
                                const float* shift, int act, float alpha, float* out, unsigned* range_acc, hipStream_t s);

// First-layer direct convolution, Cin = 3 (conv1 7x7/2 of the ResNet front, conv1_x 5x5/2 of
// the paper encoder), fused scale/shift + activation.  w_packed: [kh*kw*3][Cout].
hipError_t launch_conv_first(const float* x, int N, int H, int W, const float* w_packed, int KH,
                             int stride, int pad, int Cout, const float* scale, const float* shift,
                             int act, float alpha, float* out, int Ho, int Wo, hipStream_t s);

// The same layer for PREC_F16X3 models on the f16 matrix cores (conv1.hip), 128x128 input: 7x7/2, pad 3, 3 -> 64 (resnet50 front)
// or 5x5/2 'SAME', 3 -> 128 (paper encoder).  w_alt: split-f16 panel, conv1_f16x3_panel_floats() floats, element
// (kh, hi/lo plane, kw, c, cout) at half index conv1_f16x3_panel_index().
bool conv1_f16x3_supported(int KH, int Cout);
size_t conv1_f16x3_panel_floats(int KH, int Cout);
size_t conv1_f16x3_panel_index(int KH, int kh, int plane, int kw, int c, int cout);
// Mixed-object batches: samples [start[g], start[g+1]) use panel g (one launch for the whole batch).
struct Conv1Groups {
    int n_groups;
    int start[IGEMM_MAX_GROUPS + 1];
    const float* w[IGEMM_MAX_GROUPS];
    const float* scale[IGEMM_MAX_GROUPS];
    const float* shift[IGEMM_MAX_GROUPS];
};
// The split-K reduction of a mixed-object batch: rows [start[g], start[g + 1]) take object g's scale / shift (G.w unused).
hipError_t launch_splitk_reduce_groups(const float* partial, int ksplit, int M, int Cout, const Conv1Groups& G, int act, float alpha, float* out,
                                       unsigned* range_acc, hipStream_t s);
hipError_t launch_conv1_f16x3(const float* x, int N, int KH, int Cout, const Conv1Groups& G, int act, float alpha, float* out, float* pool_out,
                              unsigned* range_acc, hipStream_t s);

// MaxPooling2D 3x3 stride 2, TF 'SAME' (pad 0 before / 1 after), NHWC, C % 4 == 0.
hipError_t launch_maxpool3s2(const float* x, int N, int H, int W, int C, float* out, hipStream_t s);

// [n,128,128,4] -> xyz [n,128,128,3], prob [n,128,128,1]
hipError_t launch_split_xyzp(const float* xyzp, int64_t npix, float* xyz, float* prob, hipStream_t s);

}  // namespace p2p
