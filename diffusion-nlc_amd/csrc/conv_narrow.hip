// 3x3 stride-1 pad-1 convolution with at most 16 output channels (bf16 / f16): the networks' last layer, GroupNorm -> SiLU -> conv to
// 6 / 3 channels (src/unet_adm.py:613-617, src/unet_simple.py conv_out).  The 128-channel N-tile of conv_halo.hip did 21x the
// matrix work for it and - one barrier, one counted wait and 2.7 DMA pieces per wave for every tap - was issue-bound at 0.76 us per
// k-step (436 us per launch on 256 -> 6 @256^2, B = 16, for 537 MB of input).
//
// Here a workgroup owns a 16 x 16 pixel patch and ALL output channels (one 16-row MFMA N-tile, the rows beyond Cout are zero
// weights).  Per 64-channel block it brings the 18 x 18 input halo (41 KiB) AND the block's 18 weight fragments (9 taps x 2 k-halves,
// 16 rows x 64 B each) into LDS, then runs the nine taps straight through: 36 MFMAs per wave and ONE wait + barrier per channel
// block instead of nine.  Two stages, the next block's DMA flies under this block's MFMAs.  Persistent over the tile list.  The
// kernel is bound by the halo traffic (1.27 x the input from HBM).
//   512 threads = 8 waves; wave w owns patch rows 2 w, 2 w + 1 (two M-tiles of 16 pixels).  Operands swapped as in conv_halo.hip:
//   D[channel][pixel], lane (fr, fq) holds channels fq * 4 + reg of pixel fr.  Halo rows are 128 B, chunk XOR-swizzled with
//   (row & 7) on the DMA source side (conv_halo.hip); weight fragment rows are 64 B.
// Shapes: bf16, Cout <= 16, one input segment with C0 % 64 == 0, H % 16 == 0, W % 16 == 0, bias only (no embedding / residual /
// activation / statistics / GroupNorm prologue / fused upsample); NHWC bf16 or NCHW f32 output.
#include "common.h"
#include "conv_params.h"

namespace {

__device__ uint4 g_zero_page_n[1024];           // 16 KiB of zeros: out-of-image halo rows read it at offset cb * 128 B (cb < 128)

constexpr int NTH = 512;
constexpr int PATCH = 16, HALO = PATCH + 2, HALO_ROWS = HALO * HALO;      // 18 x 18 = 324
constexpr int A_INSTR = 41;                     // DMA wave-instructions per halo (8 rows x 128 B each): 328 rows
constexpr int A_STAGE = A_INSTR * 8 * KB_BYTES; // 41 KiB
constexpr int WF = 18;                          // weight fragments per channel block: tap * 2 + k-half, 16 rows x 64 B = 1 KiB each
constexpr int W_STAGE = WF * 1024;              // 18 KiB
constexpr int NA = 6, NW = 3;                   // DMA wave-instructions per wave and stage: halo / weights (padding ones land in scratch)
constexpr int SCRATCH = 8 * 1024;
constexpr int NARROW_LDS = 2 * (A_STAGE + W_STAGE) + SCRATCH;             // 129,024 B

__device__ __forceinline__ void nglds(const void* gptr, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gptr), "s"(__builtin_amdgcn_readfirstlane(lds_base))
                 : "memory");
}

struct TileN { int tb, y0, x0; };

template <typename T>
__global__ __launch_bounds__(NTH, 1) void conv_narrow_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = 2, PER = 8, KBE = 64;
    const int tiles_x = p.Wout / PATCH, tiles_y = p.Hout / PATCH;
    const int ntile = p.B * tiles_y * tiles_x;
    auto decode = [&](int id) {
        const int tb = id / (tiles_y * tiles_x);
        const int trem = id - tb * tiles_y * tiles_x;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        return TileN{tb, ty * PATCH, tx * PATCH};
    };
    int tl = blockIdx.x;                             // dispatch: gridDim.x <= ntile
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const unsigned ldsW = lds0 + 2 * A_STAGE;
    const unsigned ldsScratch = ldsW + 2 * W_STAGE + wave * 1024;
    const int ncb = p.C0 / KBE;

    // ---- halo: DMA instruction q = wave + 8 j covers LDS rows 8 q .. 8 q + 7 (lane: row lane >> 3, 16-byte slot lane & 7; the
    //      source chunk is slot ^ row so that LDS chunk c of row R sits at slot c ^ (R & 7))
    const int lrow = lane >> 3, lslot = lane & 7;
    const int hchunk = lslot ^ lrow;
    const char* zero = reinterpret_cast<const char*>(g_zero_page_n);
    const char* haddr[NA];
    auto halo_addr = [&](const TileN& t) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int R = (wave + 8 * j) * 8 + lrow;
            const int hy = R / HALO, hx = R - hy * HALO;
            const int iy = t.y0 + hy - 1, ix = t.x0 + hx - 1;
            const bool ok = R < HALO_ROWS && iy >= 0 && iy < p.Hout && ix >= 0 && ix < p.Wout;
            const int64_t pixel = ((int64_t)t.tb * p.Hin + iy) * p.Win + ix;
            haddr[j] = ok ? p.x0 + (pixel * p.C0 + hchunk * PER) * ES : zero;
        }
    };
    // ---- weights: fragment f = tap * 2 + kh of channel block cb = output rows 0..15 x 32 channels; DMA instruction = one fragment
    //      (lane: row lane >> 2, slot lane & 3).  Packed layout [Cout_pad][9][Cin_pad], Cout_pad >= 16 with zero rows beyond Cout.
    const int wrow_l = lane >> 2, wslot = lane & 3;
    const int64_t wrow = (int64_t)9 * p.Cin_pad * ES;
    const char* wlane = p.w + (int64_t)wrow_l * wrow + wslot * PER * ES;
    auto issue = [&](int cb, int stage) {
        const unsigned abase = lds0 + stage * A_STAGE + wave * 8 * KB_BYTES;
        const int64_t off = (int64_t)cb * (KBE * ES);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const bool real = (wave + 8 * j) < A_INSTR;                       // wave-uniform
            nglds(haddr[j] + off, real ? abase + j * 64 * KB_BYTES : ldsScratch);
        }
        const unsigned wbase = ldsW + stage * W_STAGE;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int f = wave + 8 * j;                                       // wave-uniform
            const bool real = f < WF;
            const int fc = real ? f : 0;
            const int tap = fc >> 1, kh = fc & 1;
            nglds(wlane + ((int64_t)tap * p.Cin_pad + cb * KBE + kh * 32) * ES, real ? wbase + f * 1024 : ldsScratch);
        }
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int a_lane = wave * 2 * HALO + fr;         // halo row of (patch row 2 wave, column fr) for tap (0, 0)
    int aoffm[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) aoffm[m] = a_lane * KB_BYTES + ((fq ^ ((a_lane + m) & 7)) << 4);
    const int woff = fr * 64 + fq * 16;              // this lane's 16 bytes of a weight fragment (row fr, channels fq * 8 ..)

    const float* bias = p.bias;
    float cb4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cb4[r] = (bias && fq * 4 + r < p.Cout) ? bias[fq * 4 + r] : 0.f;

    if (tl >= ntile) return;
    TileN cur = decode(tl);
    halo_addr(cur);
    issue(0, 0);
    int stage = 0;
    for (;;) {
        const int nxt_id = tl + (int)gridDim.x;
        const bool has_next = nxt_id < ntile;
        f32x4_t acc[2] = {f32x4_t{cb4[0], cb4[1], cb4[2], cb4[3]}, f32x4_t{cb4[0], cb4[1], cb4[2], cb4[3]}};
        for (int cb = 0; cb < ncb; ++cb) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this stage has landed (own pieces) ...
            __syncthreads();                                                  // ... everybody's; the other stage is free again
            const bool last_cb = cb + 1 == ncb;
            if (!last_cb) issue(cb + 1, stage ^ 1);
            else if (has_next) { halo_addr(decode(nxt_id)); issue(0, stage ^ 1); }
            const char* As = smem + stage * A_STAGE;
            const char* Ws = smem + 2 * A_STAGE + stage * W_STAGE + woff;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int r = tap / 3, s = tap % 3;
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    const uint4 wf = *reinterpret_cast<const uint4*>(Ws + (tap * 2 + kh) * 1024);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int k = (i + r) * HALO + s;                     // compile-time
                        // channels kh * 32 + fq * 8 .. of the 64-channel block = 16-byte chunk kh * 4 + fq: (fq ^ row) ^ (kh * 4) << 4
                        const uint4 af = *reinterpret_cast<const uint4*>(As + (aoffm[k & 7] ^ (kh * 64)) + k * KB_BYTES);
                        Mfma16<T>::run(wf, af, acc[i]);
                    }
                }
            }
            stage ^= 1;
        }
        // ---- epilogue: lane (fr, fq) holds channels fq * 4 + reg of pixel (row 2 wave + i, column fr)
        const int HWo = p.Hout * p.Wout;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = cur.y0 + wave * 2 + i, x = cur.x0 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = fq * 4 + r;
                if (ch < p.Cout) {
                    const float v = acc[i][r] * p.out_scale;
                    if (p.out_mode == NLC_OUT_NHWC)
                        ElemTraits<T>::store(reinterpret_cast<T*>(p.out) + (((int64_t)cur.tb * p.Hout + y) * p.Wout + x) * p.Cout + ch, v);
                    else
                        reinterpret_cast<float*>(p.out)[((int64_t)cur.tb * p.Cout + ch) * HWo + (int64_t)y * p.Wout + x] = v;
                }
            }
        }
        if (!has_next) break;
        cur = decode(nxt_id);
        tl = nxt_id;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

bool narrow_eligible(const KParams& p, int dtype) {
    if (!nlc_is16(dtype) || p.policy != NLC_CONV_AUTO || (p.tuning & 512)) return false;       // tuning bit 9: A/B switch
    if (!(p.KH == 3 && p.KW == 3 && p.pad_t == 1 && p.pad_l == 1 && p.stride == 1) || p.ups) return false;
    if (p.Cout > 16 || p.Cout_pad < 16 || p.C1 != 0 || p.C0 % 64 || p.C0 / 64 > 128) return false;
    if (p.Hout % PATCH || p.Wout % PATCH || p.Hout != p.Hin || p.Wout != p.Win) return false;
    if (p.emb || p.res || p.act != NLC_ACT_NONE || p.gn_coef) return false;
    if ((int64_t)p.B * p.Hout * p.Wout >= (1ll << 31)) return false;
    return p.B * (p.Hout / PATCH) * (p.Wout / PATCH) >= 64;
}

}  // namespace

int nlc_conv_narrow_ok(const KParams& p, int dtype) { return narrow_eligible(p, dtype) ? 1 : 0; }

int nlc_conv_narrow_dispatch(const KParams& p, int dtype, hipStream_t stream) {
    if (!narrow_eligible(p, dtype) || p.stats) return NLC_EUNSUPPORTED;
    static DeviceOnce once;
    const int slot = nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_narrow_kernel<bf16_raw>), hipFuncAttributeMaxDynamicSharedMemorySize, NARROW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_narrow_kernel<f16_raw>), hipFuncAttributeMaxDynamicSharedMemorySize, NARROW_LDS);
    });
    const int ncu = once.ncu[slot];
    const int ntile = p.B * (p.Hout / PATCH) * (p.Wout / PATCH);
    const int grid = ntile < ncu ? ntile : ncu;
    NLC_SWITCH_16(dtype, hipLaunchKernelGGL(conv_narrow_kernel<T16>, dim3(grid), dim3(NTH), NARROW_LDS, stream, p));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { nlc_set_error("nlc_conv2d(narrow): launch failed: %s", hipGetErrorString(e)); return NLC_ELAUNCH; }
    return NLC_OK;
}
