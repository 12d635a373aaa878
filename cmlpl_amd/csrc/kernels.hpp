// Host-side launcher declarations (one per kernel family).  All launchers are
// asynchronous on `st` and return hipError_t (hipSuccess = 0).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <atomic>

#include "common.hpp"

namespace cmlpl {

constexpr int PACK_CONV = 9 * 64 * 64;       // elements of one 3x3 weight set
// Packed (per network) weights, refreshed by the optimizer every step:
//   four split-bf16 fragment sets for the 3x3 convolutions on v_mfma_f32_32x32x16_bf16 (conv3x3.hip, "fp32 on the
//   bf16 MFMA"): conv1 fwd, conv1 dgrad, conv2 fwd, conv2 dgrad; each weight as three bf16 pieces w = w1 + w2 + w3
//   (exact, see b3_split), laid out as ready-made B fragments
//     [tap][k16 step (4)][piece (3)][n tile (2)][lane (64)][8 bf16]   lane = (n & 31) + 32 * ((k >> 3) & 1), j = k & 7
//   so a tap is one linear 24 KiB copy into LDS and a fragment one conflict-free ds_read_b128
//     forward: k = ci, n = co            dgrad: k = co, n = ci, tap flipped (transposed convolution);
//   (the fused forward's tail reads the conv2 forward set too, as 16x16x32 fragments: a lane's 8 consecutive k of one n
//   are contiguous in this layout whatever the MFMA shape; the fp32 16x16x4 fragment copy it used through round 3a is
//   gone, its region of PACK_CONV floats is kept and zero);
//   k-major copies of the two "thin" weights, so that their MFMA B fragments (fixed k, 32 consecutive outputs) are
//   coalesced 128-B global reads:   w0T [Cp][64] = conv0.weight^T (Cp = C rounded up to even, pad row zero),
//                                   (wsT [bands][1024] = feat_spe.weight^T: region kept, no longer written or read --
//                                    the spectral kernels read the canonical tensor, see dense.hip)
constexpr int PACK_B3 = PACK_CONV * 3 / 2;   // floats occupied by one split weight set
constexpr int PACK_PER_NET = 4 * PACK_B3 + PACK_CONV;
__host__ __device__ inline long long pack_off_b3(int /*C*/, int /*bands*/, int which) { return (long long)which * PACK_B3; }
__host__ __device__ inline long long pack_off_frag() { return 4LL * PACK_B3; }
// bf16 element index of (tap, k, n, piece) inside one split weight set
__host__ __device__ inline int conv_b3_index(int tap, int k, int n, int p) {
  return ((((tap * 4 + (k >> 4)) * 3 + p) * 2 + (n >> 5)) * 64 + ((k >> 3) & 1) * 32 + (n & 31)) * 8 + (k & 7);
}
__host__ __device__ inline long long pack_off_w0t() { return PACK_PER_NET; }
__host__ __device__ inline long long pack_off_wst(int C) { return PACK_PER_NET + (long long)((C + 1) & ~1) * 64; }
// ... and conv0.weight as split-bf16 B fragments [k16 step][piece][n tile][lane][8 bf16] (= conv_b3_index with tap 0,
// k = band, n = co; bands beyond C are zero) for the conv0 stage of the fused forward
__host__ __device__ inline long long pack_off_w0b3(int C, int bands) { return pack_off_wst(C) + (long long)bands * 1024; }
__host__ __device__ inline long long pack_off_w0b3_end(int C, int bands) { return pack_off_w0b3(C, bands) + (long long)((C + 15) / 16) * 1536; }
// ... and the 3x3 weights (conv1 forward / data gradient, conv2 forward / data gradient: `which` as for the split-bf16
// sets) as TWO fp16 pieces for the three-MFMA product
// (conv3x3.hip "fp32 as two fp16 pieces"): w 2^H2_WEXP = g1 + g2, g1 = fp16(w 2^H2_WEXP), g2 = fp16 of the residual
// (unscaled: the 2^-11 of the cross terms lives in the operands), fragments [tap][k16 step][piece][n tile][lane][8 fp16];
// behind the two sets one flag word per network: != 0 when a weight left fp16's range at that scale (set by the packing
// kernels, sticky until the next full pack) -- the kernels then take the three-piece bf16 loop
constexpr int H2_WEXP = 13;                 // |w| < 8 keeps w 2^13 under fp16's 65504
constexpr int PACK_H2 = PACK_CONV;          // floats occupied by one two-piece fp16 set (2 pieces x 2 bytes per weight)
__host__ __device__ inline long long pack_off_h2(int C, int bands, int which) { return pack_off_w0b3_end(C, bands) + (long long)which * PACK_H2; }
__host__ __device__ inline long long pack_off_h2flag(int C, int bands) { return pack_off_w0b3_end(C, bands) + 4LL * PACK_H2; }
__host__ __device__ inline long long pack_total(int C, int bands) { return pack_off_h2flag(C, bands) + 16; }
// fp16 element index of (tap, k, n, piece) inside one two-piece set
__host__ __device__ inline int conv_h2_index(int tap, int k, int n, int p) {
  return ((((tap * 4 + (k >> 4)) * 2 + p) * 2 + (n >> 5)) * 64 + ((k >> 3) & 1) * 32 + (n & 31)) * 8 + (k & 7);
}
struct PackInfo { long long stride, off_w0, off_w1, off_w2, off_ws; int C, bands; };
constexpr int PART3 = 9 * 4096 + 64;         // conv3x3 wgrad partial: dW[s][ci][co] + db[co]
constexpr size_t LDS_MAX = 160 * 1024;

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per
// DEVICE (the attribute lives with the device's code object): a per-call-site bit mask indexed by hipGetDevice().
struct DevOnce { std::atomic<uint64_t> mask{0}; };
template <class... Ks>
inline hipError_t ensure_max_lds(DevOnce& once, Ks... kernels) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (once.mask.load(std::memory_order_acquire) & bit) return hipSuccess;
  const void* ks[] = {(const void*)kernels...};
  for (const void* k : ks)
    if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX)) != hipSuccess) return e;
  once.mask.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}

// ---- augment.hip
// which: bit 0 = the patches (xn), bit 1 = the spectra (sn, snT)
hipError_t launch_augment(int which, int nets, int bt, int btu, int per_xp, int per_x, int lab0, int unl_base,
                          const float* xpl, const float* xl, const float* xpu, const float* xu,
                          const float* const* noise8, float sigma, uint64_t seed, uint64_t step,
                          float* xn, float* sn, float* snT, hipStream_t st,
                          const long long* labels = nullptr, float* labels_f = nullptr, const RowSel* sel = nullptr);
hipError_t launch_dist_unpack(const float* recv_f, const float* recv_z, int W, int bt_l, int btu_l, int K, float* logits_g,
                              float* feat_g, long long* labels_g, hipStream_t st);

hipError_t launch_extract_patches(const float* cube, int rows, int cols, int C, int w, const long long* idx, int n,
                                  float* out, hipStream_t st);

// ---- conv3x3.hip
hipError_t launch_pack_weights(int nets, const float* params, long long pstride, const PackInfo& pi, float* packed,
                               hipStream_t st);
struct Conv3Plan { int S, MTW; size_t lds; int nw; int ks; };   // nw: waves of the workgroup (4, or 8: one workgroup per CU, several tiles per wave); ks: the barrier-free tap loop (S = 1, one tile per wave)
bool plan_conv3(int mode, int H, int W, int rows, Conv3Plan* p);
// mode 0: out = avgpool2(relu(conv(in)+bias+in)), mask_out = relu bits; in [nets][n][H*W][64]
// mode 1: in = dpool [nets][n][(H/2)*(W/2)][64] + mask_in; out = dgrad(dz) + dz, [nets][n][H*W][64]
// what a general 3x3 launch needs for the two-piece tap loop: this map's two-piece weight set (pack_off_h2), the
// networks' range flags, the per-sample statistics table ([4][2][n], or null) and which of its rows this launch's image is
struct Conv3H2 { const float* wpk16; long long wpk16_ns; const uint32_t* h2flag; uint32_t* hstat; int kind; };
bool conv3_h2x_general(int mode, int H, int W, int rows);
hipError_t launch_conv3(int mode, int nets, int n, int H, int W, const float* in, const uint8_t* mask_in,
                        const float* wpk, long long wpk_nstride, const float* bias, long long bias_nstride,
                        float* out, uint8_t* mask_out, hipStream_t st, const Conv3H2* h2 = nullptr /* two-piece tap loop where the plan allows, or null */);
bool conv3_fused_ok(int H, int W, int C, int rows);
// the rest of the forward (conv2 + pool + head) in the same per-sample workgroup: see conv3_fwd_tail
struct FwdTail {
  const float* w2f; long long w2f_ns; const float* b2; const float* wc; const float* bc; long long p_ns;
  const float* y; const float* dropmask; float* dropgen; float* catd; float* ynorm; float* logits; float* feat;
  float* p2; uint8_t* m2; float dropout_p; int train, K;
  const float* w1h = nullptr; long long w1h_ns = 0; const uint32_t* h2flag = nullptr;   // conv1's two-piece fp16 set (pack_off_h2(.., 0)) + its flag words, or null
  uint32_t* hstat = nullptr;        // [4 kinds][2 networks][n] per-sample maxima for the two-piece weight gradient, or null
};
bool conv3_fused_tail_ok(int H, int W, int C, int rows, int K);
hipError_t launch_conv3_fused(int nets, int n, int C, int H, int W, const XSrc& xs, const float* w0t, long long w0t_ns,
                              const float* b0, long long b0_ns, float* a0out, const float* wpk, long long wpk_ns,
                              const float* bias, long long bias_ns, float* out, uint8_t* mask_out,
                              const FwdTail* tail /* or null */, hipStream_t st, float* xn_out = nullptr);
bool conv3_infer_ok(int H, int W, int C, int K);
hipError_t launch_conv3_infer(int n, int C, int H, int W, const float* cube, int crows, int ccols, long long pix0,
                              const float* w0t, const float* b0, const float* wpk, const float* bias, const FwdTail& t,
                              long long* labels_out, hipStream_t st);
bool conv3_fused_bwd_ok(int H, int W, int C, int rows);
// the head / conv2 part of the backward in the same per-sample workgroup: see conv3_bwd_head
struct BwdHead {
  const float* dlogits; const float* dfeat; const float* mask; const float* wc; long long p_ns;
  const float* y; const float* ynorm; const uint8_t* m2; const float* w2d; long long w2d_ns;
  float* dy; float* dp2; float* dp1; int K;
  const float* w1h = nullptr; long long w1h_ns = 0; const uint32_t* h2flag = nullptr;   // conv1's data-gradient two-piece fp16 set (pack_off_h2(.., 1)), or null
  uint32_t* hstat = nullptr;
};
bool conv3_fused_head_ok(int H, int W, int C, int rows, int K);
bool conv3_h2x_both(int H, int W, int C, int rows, int K);
hipError_t launch_conv3_fused_bwd(int nets, int n, int C, int H, int W, const float* dpool, const uint8_t* mask,
                                  const float* wpk, long long wpk_ns, const XSrc& xs, float* part0, long long part0_ns,
                                  const BwdHead* head /* or null */, hipStream_t st);
struct Wgrad3Plan { int RU, U, G, cspl, rsplit, UPG, b3; size_t lds; };   // rsplit > 0: row-split kernel with CPR = rsplit, UPG units per workgroup
bool plan_wgrad3(int nets, int n, int H, int W, Wgrad3Plan* p, int role = 0);   // role: 0 alone, 1 / 2 first / second map of a pair launch
bool plan_wgrad3_both(int nets, int n, int H1, int W1, int H2, int W2, bool want_pair, Wgrad3Plan* p1, Wgrad3Plan* p2,
                      bool* pair);
hipError_t launch_wgrad3(int nets, int n, int H, int W, const float* in, const float* dpool, const uint8_t* mask,
                         float* part, hipStream_t st);
hipError_t launch_wgrad3_pair(int nets, int n, int H1, int W1, const float* in1, const float* dpool1,
                              const uint8_t* mask1, float* part1, int H2, int W2, const float* in2,
                              const float* dpool2, const uint8_t* mask2, float* part2, bool* merged, hipStream_t st,
                              const uint32_t* hstat = nullptr /* this step's per-sample maxima (conv3x3.hip: [4][2][n]), or null */,
                              const uint32_t* h2flag = nullptr, long long h2flag_ns = 0);

// ---- conv0.hip
hipError_t launch_conv0_fwd(int nets, int n, int C, int HW, const float* xn, const float* w0t, long long w0t_ns,
                            const float* b, long long pstride, float* a0, hipStream_t st);
// augmentation + conv0 in one launch on the split-bf16 MFMA (general path; windows of a multiple of 8 pixels, C <= 128):
// raw rows in (xs), a0 out, the augmented rows to xn (or null); w0b3 = the packed split fragments of conv0 (pack_off_w0b3)
bool conv0a_ok(int C, int HW);
hipError_t launch_conv0a_fwd(int nets, int n, int C, int HW, const XSrc& xs, const float* w0b3, long long w0b3_ns,
                             const float* b, long long pstride, float* a0, float* xn, hipStream_t st);
int plan_conv0_wgrad_G(int n, int C, int HW);
// deterministic sum of per-workgroup weight-gradient partials, up to 3 tensors in one launch
struct ReduceProb { const float* part; float* dW; float* db; int G, PS, mode, C, blk0, el; };
struct ReduceTable { ReduceProb p[3]; int count, total_blocks; long long grad_ns;
                     int* dyn_cursor; cmlpl_dyn* dyn_table; /* graph replay: cursor advanced by 1, next row -> table[0]; or null */ };
void reduce_table_add(ReduceTable& t, const float* part, int G, int PS, int mode, int C, float* dW, float* db);
hipError_t launch_partial_reduce(int nets, const ReduceTable& t, hipStream_t st);
struct GemmTN;
hipError_t launch_reduce_gemm(int nets, const ReduceTable& t, const GemmTN& g0, const GemmTN& g1, hipStream_t st);
// a conv0 weight-gradient partial: [rows = C rounded up to 4][64 co] + 64 bias sums (rounds 1-3 kept whole 32-band tiles:
// at 103 bands a fifth of the partial bytes -- written by every sample-net, re-read by the reduce -- were padding)
__host__ __device__ inline int conv0_partial_rows(int C) { return (C + 3) & ~3; }
int conv0_partial_size(int C);
hipError_t launch_conv0_wgrad(int nets, int n, int C, int HW, const float* xn, const float* da0, float* part,
                              hipStream_t st, const uint32_t* zstat = nullptr /* per-sample maxima table: zero-gradient rows are not walked */,
                              const uint32_t* h2flag = nullptr, long long h2flag_ns = 0);

// ---- dense.hip
bool spe_fused_ok(int bands);
hipError_t launch_spe_fused(int nets, int n, int bands, const XSrc& xs, const float* w /* feat_spe.weight, canonical */,
                            const float* bias, long long p_ns, float* y, float* sn, const long long* labels,
                            float* labels_f, int bt, hipStream_t st);
hipError_t launch_spe_fwd(int nets, int n, int bands, const float* sn, const float* w, const float* b,
                          long long pstride, float* y, hipStream_t st);
// feat = y / ||y||, ynorm = ||y|| in a launch of its own (bit-identical to the per-sample forward's tail); feat_ns: per-net stride of `feat`
hipError_t launch_feat_norm(int nets, int n, const float* y, float* ynorm, float* feat, long long feat_ns, hipStream_t st);
// C[b][i][j] = scale * sum_r A[b][r][i] * B[b][r][j]  (+ optional colsum of A into bias[b][i])
struct GemmTN {
  const float* A; const float* B; float* C; float* bias;
  const float* bias_in; long long bias_in_bstride; int relu;   // epilogue: C = [relu](acc*scale + bias_in[j])
  long long a_bstride, b_bstride, c_bstride, bias_bstride;
  int lda, ldb, ldc, M, N, R, batches;
  float scale;
  int b_seg_rows = 0; long long b_seg_stride = 0;   // > 0: row r of B lives at (r / seg_rows) * seg_stride + (r % seg_rows) * ldb
};
hipError_t launch_gemm_tn(const GemmTN& g, hipStream_t st);
hipError_t launch_gemm_tn2(const GemmTN& g0, const GemmTN& g1, hipStream_t st);   // two problems, one launch

// ---- head.hip
hipError_t launch_head_fwd(int nets, int n, int HW4, int K, const float* p2, const float* y, const float* dropmask,
                           float* dropgen, float dropout_p, int train, uint64_t seed, uint64_t step,
                           int nlab, int lab0, int unl_base,
                           const float* wc, const float* bc, long long pstride,
                           float* catd, float* ynorm, float* logits, float* feat, hipStream_t st,
                           DynRef dyn = DynRef());
hipError_t launch_head_bwd(int nets, int n, int HW4, int K, const float* dlogits, const float* dfeat,
                           const float* dropmask, const float* wc, long long pstride,
                           const float* y, const float* ynorm,
                           float* dy, float* dp2, hipStream_t st);
// dy += relu'(y) * (dfeat - feat <feat, dfeat>) / ||y|| behind a head that ran with dfeat == null (bit-identical to the head with dfeat)
hipError_t launch_dy_fixup(int nets, int n, const float* y, const float* ynorm, const float* dfeat, float* dy,
                           hipStream_t st);

// ---- loss.hip   (row-sharded: see the header of loss.hip)
// Planner / path switches: environment variables, read once per process (and again only on
// cmlpl_debug_reload_switches), all of them here.
// -1 / 0 = "not set" where the comment says so; every default is the product path.
struct Switches {
  int fuse_conv0, fuse_conv0_bwd, fuse_tail, fuse_spe;        // CMLPL_FUSE_*: 0 = the unfused round-1 kernels (default 1)
  int fuse_big;                                               // CMLPL_FUSE_BIG: 0 = windows of 129 .. 256 pixels on the general kernels (default 1: eight-tile per-sample kernels)
  int conv3_nw8;                                              // CMLPL_CONV3_NW8: 0 = the general 3x3 kernels always with four waves (default 1)
  int conv0a;                                                 // CMLPL_CONV0A: 0 = augment_kernel + conv0_fwd_kernel (f32 MFMA) on the general path (default 1: one fused split-bf16 launch)
  int zero_skip;                                              // CMLPL_ZERO_SKIP: 0 = nothing is skipped (default 1: a sample whose gradient image is zero everywhere -- a row no loss term reaches through the convolutions -- skips conv2's / conv1's data-gradient loops and conv0's weight gradient in the fused backward and is left out of the two-piece weight gradients: exact)
  int bwd_pair;                                               // CMLPL_BWD_PAIR: 0 = the fused backward's workgroup b of network 1 takes sample b (default 1: sample (b + n / 2) % n, see wg_decode in conv3x3.hip)
  int f16x2;                                                  // CMLPL_F16X2: 0 = conv1's tap loops of the four-tile per-sample kernels on three bf16 pieces like every other product (default 1: TWO fp16 pieces, three MFMAs per product, wherever the operands' ranges allow; 2 / 3: in the forward / the backward kernel only; 4: as 1, but an all-zero gradient image runs the loop instead of skipping it -- a measurement aid)
  int conv3_ks;                                               // CMLPL_CONV3_KS: 0 = the general 3x3 kernels always with LDS-staged tap weights (default 1: barrier-free loop at S = 1, one tile per wave)
  int ks8;                                                    // CMLPL_KS8: eight-wave per-sample workgroups never (0) / always (1) / when the grid fits the CUs (-1)
  int conv3_s, conv0_dma, conv0_ps;                           // CMLPL_CONV3_S (0 = planner), CMLPL_CONV0_DMA (default 1), CMLPL_CONV0_PS (0 = planner)
  int wgrad3_u, wgrad3_cspl, wgrad3_r, wgrad3_ru, wgrad3_rg, wgrad3_pg1, wgrad3_pg2, wgrad3_b3, wgrad3_pair;   // CMLPL_WGRAD3_*
  int pair_wide, pair_nbw, pair_mb, pair16, pair_tall;        // CMLPL_PAIR_WIDE / _TALL (-1 = planner), _NBW / _MB (0 = planner), CMLPL_PAIR16 (default 1)
  int dfeat_lds;                                              // CMLPL_DFEAT_LDS (-1 = wherever the operands allow, 0 = never)
  int mb_fast, unsup_onewg, unsup_3l, ntx_mfma;               // CMLPL_MB_FAST, CMLPL_UNSUP_ONEWG, CMLPL_UNSUP_3L, CMLPL_NTX_MFMA (default 1)
  int ntx_ncw;                                                // CMLPL_NTX_NCW: 1 / 2 / 4 = 64 / 128 / 256 embedding columns per workgroup of the NT-Xent gradient (0 = planner)
};
inline Switches read_switches() {
    auto env = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
    Switches w;
    w.fuse_conv0 = env("CMLPL_FUSE_CONV0", 1); w.fuse_conv0_bwd = env("CMLPL_FUSE_CONV0_BWD", 1);
    w.fuse_tail = env("CMLPL_FUSE_TAIL", 1); w.fuse_spe = env("CMLPL_FUSE_SPE", 1);
    w.ks8 = env("CMLPL_KS8", -1); w.fuse_big = env("CMLPL_FUSE_BIG", 1); w.conv3_nw8 = env("CMLPL_CONV3_NW8", 1); w.conv3_ks = env("CMLPL_CONV3_KS", 1); w.conv0a = env("CMLPL_CONV0A", 1); w.f16x2 = env("CMLPL_F16X2", 1); w.bwd_pair = env("CMLPL_BWD_PAIR", 1); w.zero_skip = env("CMLPL_ZERO_SKIP", 1);
    w.conv3_s = env("CMLPL_CONV3_S", 0); w.conv0_dma = env("CMLPL_CONV0_DMA", 1); w.conv0_ps = env("CMLPL_CONV0_PS", 0);
    w.wgrad3_u = env("CMLPL_WGRAD3_U", 0); w.wgrad3_cspl = env("CMLPL_WGRAD3_CSPL", 0); w.wgrad3_r = env("CMLPL_WGRAD3_R", 1);
    w.wgrad3_ru = env("CMLPL_WGRAD3_RU", 0); w.wgrad3_rg = env("CMLPL_WGRAD3_RG", 0);
    w.wgrad3_pg1 = env("CMLPL_WGRAD3_PG1", 0); w.wgrad3_pg2 = env("CMLPL_WGRAD3_PG2", 0);
    w.wgrad3_b3 = env("CMLPL_WGRAD3_B3", 1); w.wgrad3_pair = env("CMLPL_WGRAD3_PAIR", 1);
    w.pair_wide = env("CMLPL_PAIR_WIDE", -1); w.pair_nbw = env("CMLPL_PAIR_NBW", 0); w.pair_mb = env("CMLPL_PAIR_MB", 0);
    w.pair16 = env("CMLPL_PAIR16", 1); w.pair_tall = env("CMLPL_PAIR_TALL", -1);
    w.dfeat_lds = env("CMLPL_DFEAT_LDS", -1);
    w.mb_fast = env("CMLPL_MB_FAST", 1); w.unsup_onewg = env("CMLPL_UNSUP_ONEWG", 1); w.unsup_3l = env("CMLPL_UNSUP_3L", 1);
    w.ntx_mfma = env("CMLPL_NTX_MFMA", 1); w.ntx_ncw = env("CMLPL_NTX_NCW", 0);
    return w;
}
// the process's table: filled on first use; cmlpl_debug_reload_switches (tests: one process walks several settings)
// reads the environment again -- between calls, never while a launch is being planned on another thread
inline Switches& switches_table() {
  static Switches sw = read_switches();
  return sw;
}
inline const Switches& switches() { return switches_table(); }

// compute units of the current device (cached per device: the pair launch shares them between its two maps; a
// host without a device -- the library loaded for its symbols only -- plans for a full MI355X)
inline int device_cus() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int v = cache[dev & 63].load(std::memory_order_relaxed);
  if (v == 0) {
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cache[dev & 63].store(v, std::memory_order_relaxed);
  }
  return v;
}

struct LossArgs {
  const float* logits; const float* feat; const int64_t* labels;   // GLOBAL [2][n][K], [2][n][1024], [bt]  (plain mode)
  // packed mode (recv_f != null; the sharded step): the GLOBAL embeddings and labels are read where the all-gather left
  // them, rank-major blocks [W][ 2*n_l*1024 feat | bt_l labels as float ] with per-rank rows [labelled ; unlabelled] --
  // they depend on the spectral branch alone and are gathered under the convolutions; the logits stay LOCAL
  // (logits_loc [2][n_l][K], this shard's rows: only they are read -- the bank write takes the un-smoothed probabilities
  // of the other ranks' rows from the gathered probabilities, which carry them anyway)
  const float* recv_f; long long pack_f; int bt_l, btu_l;
  const float* logits_loc;
  const float* bank_f[2]; const float* bank_p[2];
  float* bank_fw[2]; float* bank_pw[2];
  int Q, ptr0, ptr1;
  int bt, btu, K, smooth;                 // global labelled / unlabelled rows
  int lab0, nlab, unl0, nunl;             // this shard's rows
  int pshard;                             // rows per shard in probs_g (= btu on one GPU)
  float adap_mask, T, alpha, w_contrast, w_mutual, pos_thr, neg_thr;
  float* scalars; float* dlogits; float* dfeat;   // local layouts: [2][nlab+nunl][K], [2][nlab+nunl][1024]
  RowSel sel;                             // labels by index (plain mode) + device-side step scalars: ptr0 / ptr1 / smooth /
                                          // adap_mask / logging row are then read from the row, `scalars` is the ring base
  float* probs_l;                         // [4][nunl][K] written by phase 1
  const float* probs_g;                   // shard-major [btu/pshard][4][pshard][K] read by phase 2
  float* dfw_part;                        // [btu][1024] partial of dfeat_w over this shard's rows
  // workspace
  float* rs_part; float* ep_part; float* Smat; float* G; float* GT; float* masks; float* rowloss;
  int ctw;                                // bank columns per partial of rs_part / ep_part (set by launch_loss_phase1: 32, or the
                                          // wide kernel's columns per wave)
};
size_t loss_ws_floats(int nlab, int nunl, int btu_g, int K, int Q);
void loss_ws_carve(LossArgs& a, float* ws);
// phase 1 = pair_exp (embeddings and banks only) + the row kernel (this shard's logits); plain mode: the row launch also
// writes the banks.  Packed mode: the bank write rides with phase 2's graph launch (it needs every rank's probabilities).
hipError_t launch_loss_phase1(const LossArgs& a, hipStream_t st);
hipError_t loss_prepare_capture();   // kernel attributes a captured step may need for the first time
hipError_t launch_loss_graph(const LossArgs& a, hipStream_t st);
hipError_t launch_loss_dfeat(const LossArgs& a, hipStream_t st);

// ---- ntxent.hip
size_t ntxent_ws_floats(int B, int D);
hipError_t launch_ntxent(const float* ei, const float* ej, int B, int D, float T, float* loss, float* gi, float* gj,
                         float* ws, hipStream_t st);

// ---- optim.hip
hipError_t launch_adam(int nets, float* params, long long pstride, const float* grads, long long gstride,
                       float* m, float* v, long long live, long long t, float lr, float b1, float b2, float eps,
                       float* packed, const PackInfo& pi, hipStream_t st, DynRef dyn = DynRef());
void adam_bias_scalars(float lr, float b1, float b2, long long t, float* step_size, float* bc2_sqrt);


// ---- memobank.hip  (loss_helper.py)
hipError_t launch_mb_select(const float* prob, const float* label, const float* low_mask, const float* high_mask, int N,
                            int Nl, int K, int* lists, int* counts, hipStream_t st);
hipError_t launch_mb_proto(const float* rep_t, int D, const int* lists, const int* counts, int N, int K, float* proto,
                           hipStream_t st);
hipError_t launch_mb_enqueue(const float* rep_t, int D, const int* lists, const int* counts, int N, int K, float* bank,
                             int* state, const int* caps, int cap_stride, hipStream_t st);
hipError_t launch_mb_push(const float* keys, int m, int D, float* bank_c, int* state_c, int cap, hipStream_t st);
hipError_t launch_mb_infonce(const float* rep, int D, const int* pool, const long long* anchor_draw, const float* pos,
                             long long pos_qstride, const float* bank_c, int cap, int head, const long long* neg_draw,
                             int Qn, int NN, float temp, float scale, float* lossq, float* ganchor, float* drep,
                             hipStream_t st);
hipError_t launch_mb_sum(const float* v, int n, float* out, hipStream_t st);
struct MbPrep {
  const float* prob_l; const float* prob_u; const float* label_l; const float* label_u;   // [Nl][K], [N-Nl][K]
  const float* low_mask; const float* high_mask; const float* rep_t;
  int N, Nl, K, D;
  int* lists; int* counts; float* proto;
  float* bank; int* state; const int* caps; int cap_stride;
  int* keys_log;            // optional [K][2]: (new keys, rows after) of this call, for the host's pointer bookkeeping
};

struct MbLoss {
  const float* rep; int N, D, K, Q, NN;
  const int* lists; const int* counts; const int* state; const float* proto;
  const float* bank; const int* caps; int cap_stride;
  const long long* anchor_draw; const long long* neg_draw;   // injected [K][Q], [K][Q*NN] by loop position, or null
  uint64_t seed, call;                                         // in-kernel draws
  const float* momentum; const int* momentum_on; float ema;    // [K][Q][D] (or null); device flag "not all zero"
  float* prototype;                                            // out [K][Q][D] when momentum is given
  float temp;
  float* lossq; float* ganchor; int* arow; float* drep; float* total;
};

hipError_t launch_mb_onepass(const MbPrep& pa, const MbLoss& la, hipStream_t st);
size_t unsup_ws_bytes(int B);
hipError_t launch_unsup(const float* predict, long long* target, const float* teacher, int B, int K, double percent,
                        float* loss, float* dpredict, void* ws, hipStream_t st);

}  // namespace cmlpl
