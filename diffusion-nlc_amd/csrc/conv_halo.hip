// 3x3 stride-1 pad-1 convolution with the input HALO tile resident in LDS (gfx950).
//
// The plain implicit-GEMM kernels (conv_fast.hip / conv_igemm.hip) stage the A operand once per tap,
// i.e. the same input pixels 9 times; measured on the 256->256 @256^2 layer that kernel is staging-
// bound (1.40 ms with the MFMAs removed vs 1.19 ms with the staging removed, 1.66 ms together).
// Here a workgroup owns a 16x16 output patch of one image x 128 output channels and, per 64-channel
// block, brings the 18x18 input halo into LDS ONCE; the nine taps then read shifted windows of it.
// Only the weights are staged per tap.  LDS-DMA bytes per MFMA drop 3.1x.
//
//   512 threads = 8 waves as 4 (M) x 2 (N); each wave 64 pixels (4 patch rows) x 64 cout = 4x4 MFMA tiles.
//   LDS: 2 halo stages (328 rows x 128 B, 324 used) + 4 weight stages (128 x 128 B) + DMA scratch = 154 KiB.
//   Halo of channel block cb+1 is fetched while block cb computes; weights run THREE k-steps ahead so that
//   the fragments of the next k-step can be read from LDS while the current one is still in the MFMA pipe
//   (register double buffering across the barrier - with one barrier per k-step and all waves in lockstep
//   the MFMA pipe otherwise idles through every fragment-read burst).  One counted s_waitcnt vmcnt(N) +
//   barrier per k-step.  All DMA is issued from inline asm (see conv_fast.hip for why).
//   PERSISTENT: one workgroup per CU walks a list of tiles; the halo / weight DMA streams run straight across tile
//   boundaries (the next tile's first halo + weights land under the current tile's last k-steps) and the epilogue
//   writes the accumulators to global memory straight from registers (MFMA operands swapped so that a lane holds
//   16 consecutive channels of one pixel) - no LDS transpose, no barrier, asynchronous stores.  Before this a
//   workgroup paid ~45 % of its time in the per-tile prologue + LDS-staged epilogue with nothing to overlap them
//   (1 workgroup per CU).
//   Rows are XOR-swizzled chunk ^ (row & 7): conflict-free ds_read_b128 for 16 consecutive rows at ANY
//   alignment, which the tap shifts need (brute-forced).
#include "common.h"
#include "conv_params.h"

namespace {

__device__ uint4 g_zero_page_h[1024];            // 16 KiB of zeros: out-of-image halo rows read it at offset cb * 128 B (cb < 128)

constexpr int HT = 512;                         // threads
constexpr int PATCH = 16;                       // output patch edge
constexpr int HALO = PATCH + 2;                 // 18
constexpr int HALO_ROWS = HALO * HALO;          // 324
constexpr int A_INSTR = 41;                     // DMA wave-instructions per halo (8 rows each): 328 rows
constexpr int A_STAGE = A_INSTR * 8 * KB_BYTES; // 41 KiB
constexpr int B_STAGE = BN * KB_BYTES;          // 16 KiB
constexpr int NBST = 4;                         // weight stages (3 steps ahead)
constexpr int NA = 6, NB = 2;                   // DMA wave-instructions per wave: per halo / per weight tile
constexpr int SCRATCH = 8 * 8 * KB_BYTES;       // landing zone of the padding DMA instructions (8 KiB)
constexpr int COEF_STAGE = 1024;                // GroupNorm prologue: (a, b) of 128 input channels = one DMA piece, per halo stage
constexpr int HALO_LDS = 2 * A_STAGE + NBST * B_STAGE + SCRATCH + 2 * COEF_STAGE;   // 159,744 B

template <typename T> using MmaH = Mfma16<T>;

__device__ __forceinline__ int hoff(int row, int chunk) { return row * KB_BYTES + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ void glds16h(const void* gptr, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gptr), "s"(__builtin_amdgcn_readfirstlane(lds_base))   // wave-uniform by construction
                 : "memory");
}
// SGPR-base form: address = sbase (uniform, 64-bit) + voff (per lane, 32-bit unsigned) - no per-lane 64-bit address
// arithmetic in the k-loop (the VGPR-address form cost ~12 VALU incl. quarter-rate 64-bit multiplies per weight DMA)
__device__ __forceinline__ void glds16h_s(unsigned voff, const void* sbase, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(__builtin_amdgcn_readfirstlane(lds_base)), "s"(sbase)
                 : "memory");
}
template <int N> __device__ __forceinline__ void dma_wait_h() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct TileH { int tb, y0, x0, n0, cb0, cb1, ks, tix; };     // cb0 / cb1 / ks / tix: channel-block range, split index, tile index (SPLIT only)

// SPLIT: split-K for launches with fewer tiles than CUs (the 16x16 level): a list entry is (tile, channel-block range); every
// workgroup writes its accumulators as raw f32 partial sums [split][M][Cout], and the workgroup that ARRIVES LAST at a tile (one
// atomic counter per tile, self-resetting) reads all of the tile's partials back in split order - a fixed summation order whoever
// is last - adds bias / embedding and runs the normal epilogue.  No reduce pass, statistics in the usual four-per-patch form.
// A separate instantiation: the plain kernel keeps its code.
// X3 (T = float only, NLC_MATH_F16X3): the k-loop of the 16-bit kernel on f32 tensors.  A 128-byte k-block is 32 f32 channels; every
// landed halo row is rewritten IN PLACE in LDS, once per element, by the wave that DMA'd it (the GroupNorm prologue's slot in the
// schedule): 16-byte chunk c = 0..3 of a row then holds the f16 `hi` halves of channels 4c..4c+3 and 16+4c..16+4c+3, chunk 4+c the
// `lo` halves of the same eight channels (conv_params.h: f16x3_split4; the weights arrive packed that way).  The two fragment reads
// of a k-step that used to fetch the two k-halves now fetch a lane's eight hi and its eight lo values, and a k-step is
// hi*hi, hi*lo, lo*hi = 48 v_mfma_f32_16x16x32_f16 per wave instead of 64 exact-f32 MFMAs at 1/8 the rate each.
// CF (16-bit, no SPLIT / GN / X3; chosen by the dispatch): whole 128-channel N-tiles and a 16-byte-aligned bias (and embedding, if any) -
// the next tile's initial accumulator values are fetched by explicit vector loads at the top of the epilogue and waited for with a
// count of this path's own stores (see the epilogue).  A separate instantiation so that no other variant's loads share its registers.
template <typename T, bool GN, bool SPLIT = false, bool X3 = false, bool CF = false>
__global__ __launch_bounds__(HT, 2) void conv_halo_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PER = ElemTraits<T>::kPerChunk;
    constexpr int KBE = MmaH<T>::KBE;
    constexpr int ES = (int)sizeof(T);

    // ---- persistent tile schedule: XCD x (= workgroup id & 7, the dispatcher's round-robin) owns a contiguous
    //      chunk of the tile list (N-tile fastest, so neighbours in time share the halo and the weights in that
    //      XCD's L2); the workgroups of an XCD walk their chunk with stride (workgroups per XCD).
    const int tiles_x = p.Wout / PATCH, tiles_y = p.Hout / PATCH;          // (Hout, Wout) = 2 x (Hin, Win) with the fused nearest-2x upsample
    const int ksp = SPLIT ? p.ksplit : 1;
    const int nblk = p.B * tiles_y * tiles_x * p.NT * ksp;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
    const int gx = (G - xcd + 7) >> 3;
    const int cq = nblk >> 3, cr = nblk & 7;
    const int chunk_start = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_len = cq + (xcd < cr ? 1 : 0);
    auto decode = [&](int tl) {
        int id = chunk_start + tl;
        int ks = 0, cb0 = 0, cb1 = 0, tix = 0;
        if constexpr (SPLIT) {
            const int nc = p.Cin_pad / KBE;
            const int q = id / ksp;
            ks = id - q * ksp; id = q; tix = q;
            cb0 = (nc * ks) / ksp; cb1 = (nc * (ks + 1)) / ksp;
        }
        const int mt = id / p.NT, nt = id - mt * p.NT;
        const int tb = mt / (tiles_y * tiles_x);
        const int trem = mt - tb * tiles_y * tiles_x;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        return TileH{tb, ty * PATCH, tx * PATCH, nt * BN, cb0, cb1, ks, tix};
    };
    int tl = wi;
    if (tl >= chunk_len) return;                     // workgroup-uniform
    TileH cur = decode(tl);
    // A workgroup's next list entry is always `gx` entries further on: instead of decoding it (three dependent integer divisions by
    // run-time divisors, ~0.7 k cycles of a k-step with both waves of a SIMD in the same scalar code - tools/halo_stamps.py
    // STAMP_STEPS=1, tap 4 of a tile's first block) the un-split instantiations ADD the stride, decomposed once into the list's mixed
    // radix (N-tile fastest, then patch column, patch row, image), with one conditional subtraction per digit.
    int adv_n0 = 0, adv_x0 = 0, adv_y0 = 0, adv_tb = 0;
    if constexpr (!SPLIT) {
        const int c1 = gx / p.NT, c2 = c1 / tiles_x;
        adv_n0 = (gx - c1 * p.NT) * BN;
        adv_x0 = (c1 - c2 * tiles_x) * PATCH;
        adv_tb = c2 / tiles_y;
        adv_y0 = (c2 - adv_tb * tiles_y) * PATCH;
    }
    auto advance = [&](const TileH& t) {             // the entry gx further on (un-split lists; wave-uniform)
        TileH r = t;
        r.n0 += adv_n0;
        int carry = r.n0 >= p.NT * BN ? 1 : 0;
        r.n0 -= carry ? p.NT * BN : 0;
        r.x0 += adv_x0 + carry * PATCH;
        carry = r.x0 >= p.Wout ? 1 : 0;
        r.x0 -= carry ? p.Wout : 0;
        r.y0 += adv_y0 + carry * PATCH;
        carry = r.y0 >= p.Hout ? 1 : 0;
        r.y0 -= carry ? p.Hout : 0;
        r.tb += adv_tb + carry;
        return r;
    };

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lrow = lane >> 3;                      // row within a DMA wave-instruction (8 rows x 128 B)
    const int lslot = lane & 7;                      // LDS 16-byte slot this lane's DMA lands in
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const unsigned ldsB = lds0 + 2 * A_STAGE;
    const unsigned ldsScratch = ldsB + NBST * B_STAGE + wave * 8 * KB_BYTES;
    const unsigned ldsCoef = ldsB + NBST * B_STAGE + SCRATCH;
    char* smemB = smem + 2 * A_STAGE;
    const char* smemCoef = smemB + NBST * B_STAGE + SCRATCH;
    // GroupNorm prologue (bf16 only): the normalisation (+FiLM) (+SiLU) that precedes this convolution in the network is applied
    // to every landed halo row IN LDS, once per element, by the wave that DMA'd it - the separate apply pass over HBM
    // (read + write of the whole activation) disappears.  Wave 1 additionally fetches the 64 (a, b) pairs of each channel
    // block (one 1-KiB piece = 128 channels, the second half is the next block's) into a two-stage LDS table.
    constexpr bool has_gn = GN && sizeof(T) == 2;    // a separate instantiation: the plain kernel keeps its register allocation
    constexpr bool has_x3 = X3 && std::is_same<T, float>::value;
    constexpr bool has_xf = has_gn || has_x3;        // landed halo rows are transformed in LDS by the wave that fetched them
    static_assert(!(has_gn && has_x3), "GroupNorm prologue and split-f16 math are separate instantiations");
    if constexpr (has_x3) f16x3_enter();             // f32 -> f16 conversions saturate instead of overflowing to inf (conv_params.h)
    const bool coef_wave = has_gn && wave == 1;      // wave-uniform
    const int ncb = p.Cin_pad / KBE;
    const int nk = ncb * 9;

    // ---- this lane's (up to) 6 halo rows: DMA instruction q = wave + 8 j covers LDS rows 8q .. 8q+7; q >= 41 is padding.
    //      Per tile and per input segment (x0 | x1 of a concatenated input) the lane keeps the 64-bit source address of
    //      its chunk in channel block 0 of that segment - or of the zero page for out-of-image rows - so that issuing
    //      the halo of channel block cb is ONE uniform 64-bit add per instruction (the per-instruction pixel * C
    //      multiply, source select and zero-page select used to cost ~175 issue cycles per DMA, 1000+ per halo).
    //      Source-side swizzle: R & 7 == lrow.
    const char* haddr[NA];
    const char* cbase = nullptr;                     // coef_wave: the current image's coefficient rows (wave-uniform)
    unsigned hvalid = 0;                             // bit j: halo row of instruction j lies inside the image (prologue rows only)
    const int hchunk = lslot ^ lrow;
    const int cbs1 = p.C0 / KBE;                     // first channel block of the second segment (dispatch: C0 % KBE == 0)
    const char* zero = reinterpret_cast<const char*>(g_zero_page_h);
    auto halo_addr = [&](const TileH& t, int seg) {
        const char* src = seg ? p.x1 : p.x0;
        const int C = seg ? p.C1 : p.C0;
        hvalid = 0;
        if (coef_wave) cbase = reinterpret_cast<const char*>(p.gn_coef) + ((int64_t)t.tb * p.Ctot * 2) * 4;
        // pixel indices fit 32 bits (dispatch: B * Hout * Wout < 2^31); one 32 x 32 -> 64-bit multiply-add per row for the byte offset
        const unsigned pixbytes = (unsigned)C * ES;                   // wave-uniform
        const int row0 = t.tb * p.Hin;                                // wave-uniform: the image's first source row
        const unsigned choff = (unsigned)(hchunk * PER * ES);
        const int sh = p.ups ? 1 : 0;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int R = (wave + 8 * j) * 8 + lrow;
            const int hy = R / HALO, hx = R - hy * HALO;
            const int iy = t.y0 + hy - 1, ix = t.x0 + hx - 1;
            // (iy, ix) are coordinates in the conv's (possibly virtual, nearest-2x upsampled) input = output grid; with the
            // upsample fused the halo row is fetched from source pixel (iy >> 1, ix >> 1) - the LDS image holds the
            // upsampled patch, so the k-loop does not know about it (src/unet_adm.py:107-109)
            const bool ok = R < HALO_ROWS && iy >= 0 && iy < p.Hout && ix >= 0 && ix < p.Wout;
            const unsigned pixel = (unsigned)((row0 + (iy >> sh)) * p.Win + (ix >> sh));       // (garbage when !ok: not used)
            haddr[j] = ok ? src + ((uint64_t)pixel * pixbytes + choff) : zero;
            hvalid |= (ok ? 1u : 0u) << j;
        }
    };
    // The same addresses moved by ONE wave-uniform byte distance (rows that point at the zero page stay there): what a change of tile
    // comes to when the new patch touches the same image borders as the old one - source addresses are linear in (image, y, x) - and
    // what a change of input segment comes to when both segments have the same channel count.  halo_addr costs ~3.8 k cycles where it
    // runs (tap 0 of a tile's last channel block; 64-bit multiplies per row - tools/halo_stamps.py STAMP_STEPS=1); a persistent
    // workgroup's next patch is 16 patches further on in the list, i.e. mostly straight down in the same image: 13 of 16 tile changes
    // on a 256-wide map keep their border pattern.  This is six masked 64-bit adds.
    auto halo_shift = [&](int64_t delta) {
#pragma unroll
        for (int j = 0; j < NA; ++j) haddr[j] = haddr[j] != zero ? haddr[j] + delta : haddr[j];
    };
    auto same_borders = [&](const TileH& a, const TileH& b) {       // wave-uniform
        return ((a.y0 == 0) == (b.y0 == 0)) && ((a.y0 + PATCH == p.Hout) == (b.y0 + PATCH == p.Hout)) &&
               ((a.x0 == 0) == (b.x0 == 0)) && ((a.x0 + PATCH == p.Wout) == (b.x0 + PATCH == p.Wout));
    };
    auto tile_distance = [&](const TileH& a, const TileH& b, int C) -> int64_t {       // bytes from a's halo rows to b's, same segment (wave-uniform)
        const int sh = p.ups ? 1 : 0;
        const int dpix = ((b.tb - a.tb) * p.Hin + ((b.y0 - a.y0) >> sh)) * p.Win + ((b.x0 - a.x0) >> sh);      // (multiples of 16: the shifts are exact)
        return (int64_t)dpix * (int64_t)((unsigned)C * ES);
    };
    const int64_t wrow = (int64_t)9 * p.Cin_pad * ES;

    // halo instructions [J0, J0 + N) of channel block cb (of the segment haddr was set up for)
    auto issue_A = [&](int cb, int astage, auto j0_c, auto n_c) {
        constexpr int J0 = decltype(j0_c)::value, N = decltype(n_c)::value;
        const unsigned base = lds0 + astage * A_STAGE + wave * 8 * KB_BYTES;
        const int64_t off = (int64_t)(cb >= cbs1 && p.C1 > 0 ? cb - cbs1 : cb) * (KBE * ES);     // wave-uniform
#pragma unroll
        for (int j = J0; j < J0 + N; ++j) {
            const bool real = (wave + 8 * j) < A_INSTR;                       // wave-uniform
            glds16h(haddr[j] + off, real ? base + j * 64 * KB_BYTES : ldsScratch);
        }
    };
    // coefficient piece of GLOBAL channel block cb (over cat(x0, x1)) into coefficient stage `astage` (wave 1 only)
    auto issue_coef = [&](int cb, int astage) {
        unsigned voff;                               // lane * 16, recomputed here on purpose (as a loop invariant it was spilled)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(voff));
        glds16h_s(voff, cbase + (int64_t)cb * (KBE * 8), ldsCoef + astage * COEF_STAGE);
    };
    // in-place normalisation of this lane's 16 bytes of own halo instruction j (8 channels hchunk*8.. of one pixel), in two
    // parts so that the LDS read latency of the data hides under an MFMA cluster: xload() issues the read, xfinish() reads the
    // coefficients, computes and writes back.  Out-of-image rows stay zero: the convolution pads the NORMALISED input.
    auto xptr = [&](int astage, int j) { return smem + astage * A_STAGE + (wave * 8 + j * 64) * KB_BYTES + lane * 16; };
    auto xload = [&](int astage, int j) -> uint4 {
        if constexpr (has_xf) return *reinterpret_cast<const uint4*>(xptr(astage, j));
        else return uint4{0, 0, 0, 0};
    };
    auto xfinish = [&](int astage, int j, const uint4& d) {
        if constexpr (has_gn) {
            if ((hvalid >> j) & 1u) {
                const float* cf = reinterpret_cast<const float*>(smemCoef + astage * COEF_STAGE) + hchunk * 16;
                float v[8];
                chunk_to_f32<T>(d, v);
                const bool silu = p.gn_act == NLC_ACT_SILU;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 ab = *reinterpret_cast<const float4*>(cf + q * 4);       // a0 b0 a1 b1
                    float y0 = fmaf(v[2 * q], ab.x, ab.y);
                    float y1 = fmaf(v[2 * q + 1], ab.z, ab.w);
                    if (silu) { y0 = silu_f(y0); y1 = silu_f(y1); }
                    v[2 * q] = y0; v[2 * q + 1] = y1;
                }
                *reinterpret_cast<uint4*>(xptr(astage, j)) = f32_to_chunk<T>(v);
            }
        }
        if constexpr (has_x3) {
            // this lane's 16 bytes = channels 4c..4c+3 of one halo pixel, c = hchunk; the lane that holds chunk c ^ 4 of the same
            // pixel is lane ^ 4 (slot = chunk ^ (row & 7)).  Chunk c < 4 keeps both lanes' hi halves, chunk c >= 4 both lanes' lo
            // halves, each in channel order: one 8-byte exchange across the pair, VALU only (DPP row shifts by 4 under bank masks).
            uint2 hi, lo;
            f16x3_split4(d, hi, lo);
            const bool keeps_hi = (hchunk & 4) == 0;
            const uint2 send = keeps_hi ? lo : hi;
            auto xor4 = [](unsigned x) {
                int r = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0x5, false);       // row_shl:4 -> lanes 0-3, 8-11 of a row read lane + 4
                r = __builtin_amdgcn_update_dpp(r, (int)x, 0x114, 0xf, 0xa, false);           // row_shr:4 -> lanes 4-7, 12-15 read lane - 4
                return (unsigned)r;
            };
            const uint2 recv = make_uint2(xor4(send.x), xor4(send.y));
            const uint4 o = keeps_hi ? make_uint4(hi.x, hi.y, recv.x, recv.y) : make_uint4(recv.x, recv.y, lo.x, lo.y);
            *reinterpret_cast<uint4*>(xptr(astage, j)) = o;
        }
    };
    auto xform = [&](int astage, int j) { xfinish(astage, j, xload(astage, j)); };
    unsigned woff[NB];                               // per-lane byte offset of this lane's weight row + chunk (constant)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = (wave + 8 * j) * 8 + lrow;   // DMA instruction q = wave + 8 j covers LDS rows 8q..8q+7
        const int r = row & 15, jj = (row >> 4) & 3;
        const int ch = (row & 64) + (r >> 2) * 16 + jj * 4 + (r & 3);
        const int gchunk = lslot ^ (row & 7);
        woff[j] = (unsigned)((int64_t)ch * wrow + (int64_t)gchunk * PER * ES);      // < 128 * 9 * Cin_pad * ES: fits 32 bits
    }
    auto issue_B = [&](int n0, int cbb, int kt, int bstage) {       // cbb: first channel block of the tile kt counts from (0 unless SPLIT)
        const int cbl = kt / 9, tap = kt - cbl * 9;
        const int cb = cbb + cbl;
        const unsigned base = ldsB + bstage * B_STAGE + wave * 8 * KB_BYTES;
        const char* sb = p.w + (int64_t)n0 * wrow + ((int64_t)tap * p.Cin_pad + cb * KBE) * ES;     // wave-uniform
#pragma unroll
        for (int j = 0; j < NB; ++j) glds16h_s(woff[j], sb, base + j * 64 * KB_BYTES);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4_t acc[4][4];
    const int a_lane = wm * 4 * HALO + fr;           // halo row of (patch row wm*4, patch col fr) for tap (0,0)
    const int b_lane = wn * 64 + fr;

    // Per-lane LDS byte offsets of every fragment this lane will ever read, computed ONCE: the halo rows
    // (patch row + r, col + s) for the 6 x 3 (row, column) shifts and both k halves, and the weight rows.
    // With the taps unrolled at compile time the k-loop then carries no address arithmetic beyond one
    // add of the stage base per read (measured: the runtime-tap version spent more VALU issue cycles on
    // addresses than the MFMAs took).
    // Only the k-half 0 offsets are kept: chunk (4 + fq) ^ r == (fq ^ r) ^ 4, so half 1 is "^ 64".
    // Halo rows: row = a_lane + k with k = yy * 18 + s (yy = 0..5, s = 0..2) and a_lane = wm * 72 + fr, so row & 7 = (fr + k) & 7:
    // the swizzle term takes only 8 values over the 18 shifts.  Eight registers aoffm[m] = a_lane * 128 + ((fq ^ ((fr + m) & 7)) << 4)
    // and the compile-time constant k * 128 in the ds_read's immediate offset replace eighteen precomputed offsets (those, with the
    // rest of the loop-invariant addresses, no longer fitted beside the GroupNorm prologue's temporaries).
    int aoffm[8], boff;
#pragma unroll
    for (int m = 0; m < 8; ++m) aoffm[m] = a_lane * KB_BYTES + ((fq ^ ((fr + m) & 7)) << 4);
    boff = hoff(b_lane, fq);                         // + j*16 rows = + j*2048 bytes (same row & 7)

    // fragment sets (register double buffer)
    uint4 fa0[4], fb0[4], fa1[4], fb1[4];
    uint4 fa2[has_x3 ? 4 : 1], fb2[has_x3 ? 4 : 1];     // X3: the next k-step's hi fragments (this step's stay live through all three products)
    auto load_frags = [&](uint4 (&fa)[4], uint4 (&fb)[4], int astage, int bstage, auto tap_c, auto kk_c) {
        constexpr int tap = decltype(tap_c)::value, kk = decltype(kk_c)::value;
        constexpr int r = tap / 3, s = tap % 3;
        const char* As = smem + astage * A_STAGE;
        const char* Bs = smemB + bstage * B_STAGE + (boff ^ (kk * 64));
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + j * 16 * KB_BYTES);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = (i + r) * HALO + s;                                    // compile-time: i, r, s are
            fa[i] = *reinterpret_cast<const uint4*>(As + (aoffm[k & 7] ^ (kk * 64)) + k * KB_BYTES);
        }
    };
    // operands swapped: D[m = channel][n = pixel]; lane (fr, fq) holds pixel fr, channels fq*4 + reg of MFMA tile j
    auto mma16 = [&](const uint4 (&fa)[4], const uint4 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (has_x3) mfma_f16(fb[j], fa[i], acc[i][j]);
                else MmaH<T>::run(fb[j], fa[i], acc[i][j]);
            }
    };

    // ---- epilogue straight from registers: lane (fr, fq) of wave (wm, wn) holds, for patch row wm*4 + i and
    //      column fr, the 16 consecutive output channels n0 + wn*64 + fq*16 + [0, 16)  (acc[i][j][reg] -> j*4 + reg).
    //      bias + embedding are folded into the accumulators' INITIAL value (loaded for the next tile while the
    //      current one is being stored), the residual chunks are all requested before the first one is used.
    const int HWo = p.Hout * p.Wout;
    const bool vec_ok = (p.Cout % PER) == 0;
    // bf16: bias + embedding are folded into the accumulators' initial value.  f32 (the parity path) keeps the
    // reference's order instead - conv sum, then + bias, then + embedding, then + residual - so that it rounds like
    // F.conv2d(x, w, b) + emb + res does; there the accumulators start at zero.
    constexpr bool FOLD = sizeof(T) == 2;
    // Branch-free on purpose: with a per-element "if (in range) load" the compiler emitted 16 x (global_load_dword;
    // s_waitcnt vmcnt(0)) - every load paid a full memory latency AND drained the in-flight weight DMAs and stores;
    // in-kernel stamps showed the epilogue at 11-12 k cycles of a 68 k-cycle tile.  Now: clamped indices (the zero page
    // stands in for a missing bias / embedding), 32 independent loads, one wait.
    const float* zf = reinterpret_cast<const float*>(g_zero_page_h);
    auto load_cadd2 = [&](const TileH& t, float (&cb_)[16], float (&ce_)[16]) {
        const int n = t.n0 + wn * 64 + fq * 16;
        const float* bp = p.bias ? p.bias : zf;
        const float* ep = p.emb ? p.emb + (int64_t)t.tb * p.emb_stride : zf;
        const int hb = p.bias ? 1 : 0, he = p.emb ? 1 : 0, last = p.Cout - 1;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = min(n + k, last);
            const float vb = bp[hb ? c : k], ve = ep[he ? c : k];
            const bool okc = n + k < p.Cout;
            cb_[k] = okc ? vb : 0.f;
            ce_[k] = okc ? ve : 0.f;
        }
    };
    const bool bias_vec = p.bias && ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) && !p.emb;   // 4 x 16-byte loads suffice
    auto load_cadd = [&](const TileH& t, float (&cadd)[16]) {
        const int n = t.n0 + wn * 64 + fq * 16;
        if (FOLD && bias_vec && n + 16 <= p.Cout) {          // every VMEM instruction costs its wave ~60-100 issue cycles
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n + q * 4);
                cadd[q * 4] = b4.x; cadd[q * 4 + 1] = b4.y; cadd[q * 4 + 2] = b4.z; cadd[q * 4 + 3] = b4.w;
            }
            return;
        }
        float ce_[16];
        load_cadd2(t, cadd, ce_);
#pragma unroll
        for (int k = 0; k < 16; ++k) cadd[k] = FOLD ? cadd[k] + ce_[k] : 0.f;
    };
    // sum over the 16 lanes of a DPP row (= the 16 pixels fr of one quarter-wave), result in every lane; VALU only
    // (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) - a __shfl_xor is a ds_bpermute, ~100 cycles each
    // (one v_add_f32 with a DPP source operand per stage: through __builtin_amdgcn_update_dpp the compiler emitted a zeroing move,
    //  a DPP move and a packed add per stage and value - 80 instructions for the eight statistics of a tile instead of 32)
    auto row16_sum = [&](float x) {
        asm("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
        asm("v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(x));
        asm("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
        asm("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
        return x;
    };
    auto init_acc = [&](const float (&cadd)[16]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{cadd[j * 4], cadd[j * 4 + 1], cadd[j * 4 + 2], cadd[j * 4 + 3]};
    };
    // all-reduce a lane's statistics over the 16 pixel lanes (fr) of its quarter-wave in a fixed order, then every lane adds one
    // limb of the slice's totals: ONE atomic instruction per wave and tile (Stat16::emit_row, conv_params.h)
    auto emit_stats = [&](Stat16& st16, const TileH& t, int n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { st16.s[k] = row16_sum(st16.s[k]); st16.q[k] = row16_sum(st16.q[k]); }
        st16.emit_row(p.stats, t.tb, p.Cout, n, p.stats_gran, fr);
    };
    // Pins the accumulator initialisation where it is written.  Left free, the compiler sank it (and the wait for the bias loads
    // that feed it) past the loop back-edge, where its wait-count pass assumes the worst predecessor: `s_waitcnt vmcnt(0)` at the top
    // of every tile (tools/halo_stamps.py: ~5.8 k cycles on the first k-step of a tile against 1.25 k in mid-tile).
    auto pin_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
    };
    // CF: cnext = bias + embedding from the epilogue's loads.  The loads are ordinary (compiler-visible) loads: the compiler's own
    // wait-count pass then puts `s_waitcnt vmcnt(N)` in front of the FIRST instruction that touches their destination registers,
    // with N = the vector-memory operations IT issued after them (the statistics atomic; the DMA in flight is older).  Until round 5
    // they were inline-asm loads waited for by an inline-asm `s_waitcnt` with the registers as tied "+v" operands - and the register
    // allocator, which cannot know that an asm output is not there yet, put COPIES of them (v_mov_b64) in front of that wait: reads
    // of registers still in flight.  Harmless while eight row stores sat between the loads and the wait (the compiler's own store
    // hazard wait happened to precede the copies); with the deferred rows nothing did, and whenever the bias vector missed in L2 the
    // next tile started from garbage accumulators (sporadic, a whole XCD's round of tiles at a time: tests B <= 3 at 256 x 256).
    auto cadd_sum = [&](const f32x4_t (&cq)[4], const f32x4_t (&eq)[4], float (&cnext)[16]) {
        if (p.emb) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) cnext[q * 4 + r] = cq[q][r] + eq[q][r];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) cnext[q * 4 + r] = cq[q][r];
        }
    };
    // The output stores of an epilogue retire in issue order with everything else (one vmcnt): a counted wait for a weight tile that
    // was issued AFTER them waits for their acknowledgement first - measured (tools/halo_stamps.py) at ~5.8 k cycles on the first
    // k-step of every tile against 1.25 k for a step in mid-tile.  So the weight tile the next tile's step 0 would issue (its k-step 3)
    // goes out at the START of the epilogue, ahead of the stores (its stage was freed by the last step's barrier), step 0 issues none,
    // and the counted waits of steps 0 and 1 leave the eight row stores in flight: the first wait that covers them is step 2's, three
    // k-steps after they were issued.  post_ep: 2 / 1 = step 0 / 1 of a tile that follows such an epilogue (wave-uniform).
    // DEFER (the CF instantiation): the eight row stores of a tile are not issued by its epilogue at all.  A 1-KiB store instruction
    // holds its wave until the CU's store path has taken the data (64 KiB per tile from eight waves in lockstep: the epilogue ran 3.6 k
    // cycles in the first wave and 5.9 k in the last, and the first k-step of the next tile another 1.7-3.9 k over a plain step while
    // the early finishers waited at its barrier - tools/halo_stamps.py STAMP_STEPS=1; no MFMA runs meanwhile).  The packed rows stay in
    // 32 registers and go out ONE per k-step, behind that step's DMA issue, under the next tile's steps 0-7 (a tile has >= 9); the
    // last tile of a workgroup stores at once.  `pend`: 0 = nothing pending, 1 = rows pending, 2 = rows pending and ONE statistics
    // atomic sits in the queue between the bias loads and step 0's DMA.  Every counted wait of those steps allows for the rows issued
    // since the weight tile it is for: per step (issue order: weights, halo pieces, row)
    //   step 0: [atomic] W3 h h r0 | wait for W2 (older than all of them)          -> allowance + 1 (+ 1 with the atomic)
    //   step s = 1..7: ... r(s-1) W(s+3) [h h] r(s) | wait for W(s+2), issued before r(s-1)   -> allowance + 2
    //   step 8: ... r7 W11 | wait for W10                                           -> allowance + 1
    // (a second, poisoning atomic - non-finite outputs - only makes step 0's wait stricter than it needs to be).  With the rows out
    // of the way the early weight issue below has nothing left to overtake and is off in this instantiation.
    constexpr bool DEFER = CF;
    constexpr bool peel_first = DEFER;       // (dispatch: CF launches have >= 2 channel blocks, no segment boundary behind the first, NHWC output, no activation)
    constexpr bool EARLY_W = FOLD && !SPLIT && !has_xf && !DEFER;
    int pend = 0;
    uint4 pend0 = uint4{0, 0, 0, 0}, pend1 = pend0, pend2 = pend0, pend3 = pend0, pend4 = pend0, pend5 = pend0, pend6 = pend0, pend7 = pend0;   // (named: as an array they went to scratch)
    auto pend_ref = [&](auto k_c) -> uint4& {
        constexpr int k = decltype(k_c)::value;
        if constexpr (k == 0) return pend0; else if constexpr (k == 1) return pend1; else if constexpr (k == 2) return pend2; else if constexpr (k == 3) return pend3;
        else if constexpr (k == 4) return pend4; else if constexpr (k == 5) return pend5; else if constexpr (k == 6) return pend6; else return pend7;
    };
    T* pend_op = nullptr;
    auto res_ptr = [&](const TileH& t, int i) -> const T* {            // row i of this lane's four, its 16 channels
        const int n = t.n0 + wn * 64 + fq * 16;
        if (!p.res_ups) return reinterpret_cast<const T*>(p.res) + ((((int64_t)t.tb * p.Hout + t.y0 + wm * 4 + i) * p.Wout + t.x0 + fr) * p.Cout + n);
        return reinterpret_cast<const T*>(p.res) + res_row(p, t.tb, t.y0 + wm * 4 + i, t.x0 + fr) * p.Cout + n;
    };
    int post_ep = 0;
    int bcur = 0, hs = 0;                            // weight stage / halo stage of the current k-step (run across tiles)
    auto epilogue = [&](const TileH& t, const TileH& nx, bool has_next) {
        bool early = false;
        if constexpr (EARLY_W) {
            early = has_next && vec_ok && (t.n0 + BN <= p.Cout) && p.out_mode == NLC_OUT_NHWC && p.act == NLC_ACT_NONE;     // workgroup-uniform: the hot path below
            if (early) issue_B(nx.n0, 0, 3, (bcur + 3) & 3);
        }
        float cnext[16];
        // The next tile's bias vector, requested NOW so that it lands while this tile's rows are converted.  Left alone the compiler sinks
        // such loads (and their wait) to the accumulator initialisation at the end of the epilogue - the load latency in the open and, the
        // counter being in order, a drain of every DMA ahead of them.  So (CF: 16-bit, aligned bias / embedding, whole 16-channel slices):
        // four 16-byte loads per vector pinned here by an empty asm that clobbers memory, and consumed by cadd_sum (see there).
        f32x4_t cq[4], eq[4];
        constexpr bool cfast = CF;
        static_assert(!CF || (FOLD && !SPLIT && !has_xf), "CF: plain 16-bit kernel only");
        if constexpr (CF) {
            // SGPR base + per-lane byte offset (lane >> 4) * 64, recomputed here on purpose: as a loop-invariant 64-bit per-lane
            // address it was spilled, and the reload's wait drained the DMA in flight
            unsigned voff;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\t"
                         "v_lshrrev_b32 %0, 4, %0\n\tv_lshlrev_b32 %0, 6, %0" : "=v"(voff));
            const char* sb = reinterpret_cast<const char*>(p.bias + nx.n0 + wn * 64);          // wave-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) cq[q] = *reinterpret_cast<const f32x4_t*>(sb + voff + q * 16);
            if (p.emb) {
                const char* se = reinterpret_cast<const char*>(p.emb + (int64_t)nx.tb * p.emb_stride + nx.n0 + wn * 64);
#pragma unroll
                for (int q = 0; q < 4; ++q) eq[q] = *reinterpret_cast<const f32x4_t*>(se + voff + q * 16);
            }
            asm volatile("" ::: "memory");          // the loads stay HERE (requested while this tile's rows are converted), not at their use
        } else if constexpr (SPLIT) {
#pragma unroll
            for (int k = 0; k < 16; ++k) cnext[k] = 0.f;           // raw partial sums: bias / embedding are added by the last arriver
        } else {
            load_cadd(nx, cnext);                    // in flight while this tile is stored
        }
        const int n = t.n0 + wn * 64 + fq * 16;
        Stat16 st16;                                 // this lane's 16 channels over its 4 pixels (ride-along GroupNorm statistics)
        st16.zero();
        bool done = false;
        bool inited = false;                         // the vectorised path below ran and initialised the next tile's accumulators itself
        bool parked = false;                         // SPLIT: this workgroup was not the last to arrive at its tile - no output from it
        if constexpr (SPLIT) {                       // dispatch: bf16, Cout % 128 == 0
            // The partial sums cross workgroups that may sit on different XCDs.  Memory-ordering argument (gfx950; MI355X_MICROARCH.md,
            // "Workgroup dispatch, XCD placement & inter-workgroup visibility"): a CU's vector L1 is never refreshed by other CUs'
            // stores and the eight per-XCD L2s are not coherent with each other, so a hand-off needs every byte to (a) leave the
            // producer's L2 and (b) be read past the consumer's L1.  Here
            //   * every partial is stored with a relaxed AGENT-scope atomic store = `global_store_dword ... sc1`: write-through, the line
            //     is not kept in the producing XCD's L2 (sc1 / atomic stores DROP it), so the bytes are at the memory side once the
            //     store is acknowledged;
            //   * every storing wave drains its stores (`s_waitcnt vmcnt(0)` counts acknowledgements), then the workgroup barrier, then
            //     ONE lane's agent-scope atomic add on the tile's counter (atomics execute at the memory side): the add cannot overtake
            //     any store of the workgroup;
            //   * the workgroup whose add returns ksplit - 1 is last; its other waves start loading only behind a workgroup barrier
            //     that the adding wave joins after its add has returned;
            //   * every load of the partials is a relaxed agent-scope atomic load = `global_load_dword ... sc1` to registers (never
            //     `flat_`): sc1 loads bypass the L1 and are served from L2, which cannot hold a stale copy because no plain access
            //     ever touches these lines (written sc1, read sc1; the buffer is reused launch after launch in this one role).
            // This is row 1 of the guide's table of hand-offs measured with sc1 loads in place of the acquire (one lane per storing
            // workgroup signals with an agent-scope atomic add, last arriver told by the returned value, hipMalloc memory, 4-byte
            // stores and loads, ONE workgroup per CU - which this kernel is).  It is measured behaviour of this chip, not a
            // guarantee of the HIP memory model, hence tuning bit 10: the formally ordered variant (agent-scope release before the
            // add, agent-scope acquire in the last arriver before its loads), which the stress test compares bit for bit
            // (tests/test_ops_gpu.py::test_split_k_stress).  The release is what costs: it writes back the XCD's whole dirty L2
            // (+80 us per launch measured), so it is not the default.
            __shared__ int s_last;
            // layout [split][tile][wave][accumulator tile 0..15][lane][4]: one f32x4 accumulator per 16-byte transaction, 1 KiB
            // contiguous per wave-instruction (in the [M][Cout] layout every lane of a dword access touched a cache line of its own:
            // +130 us per launch; as dwords [register][lane] the hand-off took four times the vector-memory instructions), and the
            // last arriver's lane (wave, lane) finds exactly its own elements at its own offset.  Same size: 256 x 128 floats per tile.
            // Inline asm (the atomic builtins stop at 8 bytes) with the sc1 policy of the argument above on both sides; s_nop: the
            // > 8-byte store-data hazard, which the hazard recognizer cannot see inside an asm statement.
            const int ntile = nblk / ksp;
            auto pbase = [&](int sp) { return p.partial + 1024 + ((((int64_t)sp * ntile + t.tix) * 8 + wave) * 64) * 64 + lane * 4; };     // [1024 counters][partials]
            {
                float* pp = pbase(t.ks);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc1\n\ts_nop 1" :: "v"(pp + i * 1024), "v"(acc[i][j]), "n"(j * 1024) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this lane's partial stores have been acknowledged ...
            __syncthreads();                         // ... and every lane's, before the workgroup's arrival is counted
            if (tid == 0) {
                const bool fenced = (p.tuning & 1024) != 0;
                if (fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                int* cnt = reinterpret_cast<int*>(p.partial) + t.tix;
                const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == ksp - 1;
                if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // self-resetting
                if (last && fenced) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                s_last = last;
            }
            __syncthreads();
            const bool last = s_last != 0;           // workgroup-uniform
            __syncthreads();                         // s_last may be rewritten by the next tile's epilogue
            if (last) {
                float cthis[16];
                load_cadd(t, cthis);
                // per split count a straight-line read-back: the loaded registers reach their wait untouched (the compiler does not
                // know they are still being written, so no select or copy may sit between a load and the wait)
                auto reduce = [&](auto ks_c) {
                    constexpr int KS = decltype(ks_c)::value;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{cthis[j * 4], cthis[j * 4 + 1], cthis[j * 4 + 2], cthis[j * 4 + 3]};
#pragma unroll
                        for (int u0 = 0; u0 < KS; u0 += 2) {           // two splits at a time (32 registers in flight), adds in the order split 0, 1, ...
                            const bool two = u0 + 1 < KS;              // compile-time after unrolling
                            f32x4_t tq[2][4];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                if (u == 1 && !two) break;
                                const float* pp = pbase(u0 + u) + i * 4 * 256;
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(tq[u][j]) : "v"(pp), "n"(j * 1024) : "memory");
                            }
                            if (two) asm volatile("s_waitcnt vmcnt(0)" : "+v"(tq[0][0]), "+v"(tq[0][1]), "+v"(tq[0][2]), "+v"(tq[0][3]),
                                                  "+v"(tq[1][0]), "+v"(tq[1][1]), "+v"(tq[1][2]), "+v"(tq[1][3]) :: "memory");
                            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(tq[0][0]), "+v"(tq[0][1]), "+v"(tq[0][2]), "+v"(tq[0][3]) :: "memory");
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                if (u == 1 && !two) break;
#pragma unroll
                                for (int j = 0; j < 4; ++j)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) acc[i][j][r] += tq[u][j][r];
                            }
                        }
                    }
                };
                if (ksp == 2) reduce(std::integral_constant<int, 2>{});
                else if (ksp == 3) reduce(std::integral_constant<int, 3>{});
                else reduce(std::integral_constant<int, 4>{});          // (nlc_conv_halo_ksplit: at most 4)
            } else {
                parked = true; done = true;
            }
        }
        if constexpr (FOLD) {
            if (!done)
            // hot path (bf16, whole 16-channel slice, NHWC): the option switches are hoisted out of the element loops;
            // measured with stamps, the general path below spent ~9 k cycles per tile on ~19 VALU per output element
            // (no fused activation here: in the networks only the 1x1 "linear" layers carry one, and with both activations' code in this
            //  path the compiler merged three variants' registers with a shuffle per stored dword; the general path below has them)
            if (DEFER || (vec_ok && t.n0 + BN <= p.Cout && p.out_mode == NLC_OUT_NHWC && p.act == NLC_ACT_NONE)) {      // workgroup-uniform
                const bool has_res = p.res != nullptr, has_stats = p.stats != nullptr;
                const float sc = p.out_scale;
                // row i of the lane's four is one image row further down: one address per tile, then a constant stride
                const int64_t m0 = ((int64_t)t.tb * p.Hout + t.y0 + wm * 4) * p.Wout + t.x0 + fr;
                const int64_t rstride = (int64_t)p.Wout * p.Cout;
                T* const op0 = reinterpret_cast<T*>(p.out) + m0 * p.Cout + n;
                // DF: the rows stay in registers (pend0..7, stored under the next tile); with no store in the block the residual is requested
                // two rows ahead of its use (16 fewer registers beside the pending rows).  Otherwise the rows are stored here, and all eight residual
                // chunks are requested before the first store (the output may alias nothing the compiler can see, so it kept each row's
                // loads behind the previous row's stores: four exposed memory latencies per tile).
                auto hot = [&](auto df_c) {
                    constexpr bool DF = decltype(df_c)::value;
                    uint4 rq0[4], rq1[4];
                    if (has_res) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const T* rp = res_ptr(t, i);
                            rq0[i] = *reinterpret_cast<const uint4*>(rp);
                            rq1[i] = *reinterpret_cast<const uint4*>(rp + 8);
                        }
                    }
                    auto row = [&](auto i_c) {
                        constexpr int i = decltype(i_c)::value;
                        float v[16];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) v[j * 4 + reg] = acc[i][j][reg];
                        if (has_res) {
                            const uint4 r0 = rq0[i], r1 = rq1[i];
                            float rr[16];
                            chunk_to_f32<T>(r0, rr); chunk_to_f32<T>(r1, rr + 8);
#pragma unroll
                            for (int k = 0; k < 16; ++k) v[k] += rr[k];
                        }
                        if (sc != 1.0f) {
#pragma unroll
                            for (int k = 0; k < 16; ++k) v[k] *= sc;
                        }
                        const uint4 pk0 = f32_to_chunk<T>(v), pk1 = f32_to_chunk<T>(v + 8);
                        if constexpr (DF) {
                            pend_ref(std::integral_constant<int, 2 * i>{}) = pk0; pend_ref(std::integral_constant<int, 2 * i + 1>{}) = pk1;
                        } else {
                            T* op = op0 + i * rstride;
                            *reinterpret_cast<uint4*>(op) = pk0;
                            *reinterpret_cast<uint4*>(op + 8) = pk1;
                        }
                        if (has_stats) {     // of the STORED (rounded) values - what the GroupNorm that follows reads
                            st16.add_chunk<T>(0, pk0); st16.add_chunk<T>(1, pk1);
                        }
                    };
                    row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{}); row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
                    if constexpr (!DF) { if (has_stats) emit_stats(st16, t, n); }
                    // the next tile's accumulators, initialised INSIDE this straight-line block: at the common tail below the wait-count
                    // pass has to merge every epilogue variant and drains the counter (vmcnt(0): all eight store acknowledgements + the DMA)
                    if constexpr (cfast) {
                        // (the compiler's wait for the bias / embedding loads lands here: behind them this block issued its residual loads,
                        //  all consumed above, and - not DF - its eight row stores and statistics atomic)
                        cadd_sum(cq, eq, cnext);
                        if constexpr (DF) {
                            pend = has_stats ? 2 : 1;
                            pend_op = op0;
                        }
                    }
                };
                hot(std::integral_constant<bool, DEFER>{});
                init_acc(cnext);
                pin_acc();
                if constexpr (DEFER) {
                    // the statistics atomic goes out BEHIND the use of the bias loads: the compiler's wait for those then counts nothing
                    // younger (with the atomic ahead of it, in two branches - add / poison - it became vmcnt(0): the atomic's round trip)
                    asm volatile("" ::: "memory");
                    if (has_stats) emit_stats(st16, t, n);
                    return;               // (the dispatch guarantees this path's conditions: no other epilogue code in the CF instantiation)
                }
                done = true;
                inited = true;
            }
        }
        if constexpr (has_x3) {
            // the packed weights of output row n carry the factor 2^e[n] (nlc_pack_conv_weights_ex): the sums are multiplied by
            // w_scale[n] = 2^-e[n] in place - a power of two, exact - before anything is added (n + 15 < Cout_pad: in range)
            const float4* wp4 = reinterpret_cast<const float4*>(p.w_scale + n);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 wq = wp4[q];
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i][q][0] *= wq.x; acc[i][q][1] *= wq.y; acc[i][q][2] *= wq.z; acc[i][q][3] *= wq.w; }
            }
        }
        if constexpr (!FOLD) {
            // f32 tensors (exact-f32 and split-f16 math), whole N-tile, NHWC, no activation, 16-byte-aligned bias / embedding: vector
            // loads of the bias and embedding slices and of all sixteen residual chunks up front, the reference's order of additions
            // (conv sum, + bias, + embedding, + residual, x scale), four 16-byte stores per row.  The general path below did 32 scalar
            // loads with clamps per tile and kept every row's residual loads behind the previous row's stores: 11-15 k cycles of a
            // 167 k-cycle tile of the split-f16 kernel (tools/halo_stamps.py --x3).
            const bool al16 = p.bias && ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) &&
                              (!p.emb || ((reinterpret_cast<uintptr_t>(p.emb) & 15) == 0 && (p.emb_stride & 3) == 0));
            if (!done && al16 && vec_ok && t.n0 + BN <= p.Cout && p.out_mode == NLC_OUT_NHWC && p.act == NLC_ACT_NONE) {      // workgroup-uniform
                const float4* bp4 = reinterpret_cast<const float4*>(p.bias + n);
                float4 bq[4], eq4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[q] = bp4[q];
                const bool has_emb = p.emb != nullptr, has_res = p.res != nullptr;
                if (has_emb) {
                    const float4* ep4 = reinterpret_cast<const float4*>(p.emb + (int64_t)t.tb * p.emb_stride + n);
#pragma unroll
                    for (int q = 0; q < 4; ++q) eq4[q] = ep4[q];
                }
                const int64_t m0 = ((int64_t)t.tb * p.Hout + t.y0 + wm * 4) * p.Wout + t.x0 + fr;
                const int64_t rstride = (int64_t)p.Wout * p.Cout;
                const float sc = p.out_scale;
                float* const op0 = reinterpret_cast<float*>(p.out) + m0 * p.Cout + n;
#pragma unroll
                for (int h = 0; h < 2; ++h) {                        // two rows at a time: eight residual chunks in flight (32 registers)
                    float4 rq[2][4];
                    if (has_res) {
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {
                            const int i = 2 * h + ii;
                            const float4* rp = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) +
                                                                                res_row(p, t.tb, t.y0 + wm * 4 + i, t.x0 + fr) * p.Cout + n);
#pragma unroll
                            for (int q = 0; q < 4; ++q) rq[ii][q] = rp[q];
                        }
                    }
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int i = 2 * h + ii;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float4 v = float4{acc[i][q][0], acc[i][q][1], acc[i][q][2], acc[i][q][3]};
                            v.x += bq[q].x; v.y += bq[q].y; v.z += bq[q].z; v.w += bq[q].w;
                            if (has_emb) { v.x += eq4[q].x; v.y += eq4[q].y; v.z += eq4[q].z; v.w += eq4[q].w; }
                            if (has_res) { v.x += rq[ii][q].x; v.y += rq[ii][q].y; v.z += rq[ii][q].z; v.w += rq[ii][q].w; }
                            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                            *reinterpret_cast<float4*>(op0 + i * rstride + q * 4) = v;
                        }
                    }
                }
                done = true;
            }
        }
        if (!done && n < p.Cout) {
            const bool full = vec_ok && (n + 16 <= p.Cout);
            constexpr int NCH = 16 / PER;
            int64_t mrow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) mrow[i] = ((int64_t)t.tb * p.Hout + t.y0 + wm * 4 + i) * p.Wout + t.x0 + fr;
            float cbias_[16], cemb_[16];
            if constexpr (!FOLD) load_cadd2(t, cbias_, cemb_);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t m = mrow[i];
                float v[16];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) v[j * 4 + reg] = acc[i][j][reg];
                if constexpr (!FOLD) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) { v[k] += cbias_[k]; v[k] += cemb_[k]; }     // reference order: + bias, then + emb
                }
                if (p.res) {
                    const int64_t mres = res_row(p, t.tb, t.y0 + wm * 4 + i, t.x0 + fr);
                    if (full) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            float rr[PER];
                            chunk_to_f32<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.res) + mres * p.Cout + n + c * PER), rr);
#pragma unroll
                            for (int k = 0; k < PER; ++k) v[c * PER + k] += rr[k];
                        }
                    } else {
                        const T* rp = reinterpret_cast<const T*>(p.res) + mres * p.Cout + n;
#pragma unroll
                        for (int k = 0; k < 16; ++k) if (n + k < p.Cout) v[k] += ElemTraits<T>::load(rp + k);
                    }
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = apply_act(v[k] * p.out_scale, p.act);
                if (p.out_mode == NLC_OUT_NHWC) {
                    T* op = reinterpret_cast<T*>(p.out) + m * p.Cout + n;
                    if (full) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            const uint4 pk = f32_to_chunk<T>(v + c * PER);
                            *reinterpret_cast<uint4*>(op + c * PER) = pk;
                            if constexpr (sizeof(T) == 2) {
                                if (p.stats) st16.add_chunk<T>(c, pk);       // GroupNorm statistics of what was just stored (the rounded values)
                            }
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 16; ++k) if (n + k < p.Cout) ElemTraits<T>::store(op + k, v[k]);
                    }
                } else {
                    const int64_t rem = (int64_t)(t.y0 + wm * 4 + i) * p.Wout + t.x0 + fr;
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (n + k < p.Cout) reinterpret_cast<float*>(p.out)[((int64_t)t.tb * p.Cout + n + k) * HWo + rem] = v[k];
                }
            }
        }
        if (!inited) {
            if constexpr (sizeof(T) == 2) {
                if (p.stats && n + 16 <= p.Cout && !parked) emit_stats(st16, t, n);
            }
            if constexpr (cfast) cadd_sum(cq, eq, cnext);
            init_acc(cnext);
            pin_acc();
        }
        post_ep = early ? 2 : 0;
    };

    // ---- prologue (first tile only): halo of block 0, weights of steps 0..2
    {
        float c0[16];
        if constexpr (SPLIT) {
#pragma unroll
            for (int k = 0; k < 16; ++k) c0[k] = 0.f;
        } else {
            load_cadd(cur, c0);
        }
        init_acc(c0);
    }
    auto t_cb0 = [&](const TileH& t) { if constexpr (SPLIT) return t.cb0; else return 0; };
    auto t_cb1 = [&](const TileH& t) { if constexpr (SPLIT) return t.cb1; else return ncb; };
    auto t_seg = [&](const TileH& t) { return (SPLIT && p.C1 > 0 && t.cb0 >= cbs1) ? 1 : 0; };
    auto t_nk = [&](const TileH& t) { if constexpr (SPLIT) return (t.cb1 - t.cb0) * 9; else return nk; };
    halo_addr(cur, t_seg(cur));
    if (coef_wave) issue_coef(0, 0);
    issue_A(t_cb0(cur), 0, std::integral_constant<int, 0>{}, std::integral_constant<int, NA>{});
    issue_B(cur.n0, t_cb0(cur), 0, 0);
    issue_B(cur.n0, t_cb0(cur), min(1, t_nk(cur) - 1), 1);
    issue_B(cur.n0, t_cb0(cur), min(2, t_nk(cur) - 1), 2);
    dma_wait_h<NB>();                                // halo 0 + weights 0,1 landed (weights 2 may fly)
    __syncthreads();
    if (has_xf) {                                    // first halo of the launch: transform all six own instructions at once
#pragma unroll
        for (int j = 0; j < NA; ++j) xform(0, j);
        __syncthreads();
    }
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    load_frags(fa0, fb0, 0, 0, K0{}, K0{});

    // (Tried and measured slower on this kernel, 1.18 vs 1.12 ms on 256->256 @256^2: running SIMD partner waves in
    //  complementary orders by giving waves 0-3 / 4-7 their barrier at different points of one instruction stream.)
    constexpr int wdist = 3;                         // weight tiles run 3 k-steps ahead
    // The list entry after next is worked out inside the current tile's k-loop (tap 4 of its first block), not at the top of a tile
    // between the epilogue and the first MFMA: by `advance` (a dozen scalar operations), or - split lists - by a full decode (five
    // integer divisions, ~1 k cycles of dependent scalar work, of which ~0.7 k showed as extra time of that k-step).
    TileH nxt = tl + gx < chunk_len ? decode(tl + gx) : cur;
    TileH nxt2 = nxt;
    for (;;) {
        const bool has_next = tl + gx < chunk_len;
        int kt = 0;
        const int c_end = t_cb1(cur), nkt = t_nk(cur);
        // One channel block = nine k-steps.  FIRST (DEFER only): the copy of the body that a tile's first block runs when the previous
        // tile left its rows pending - it is never the tile's last block and has no segment boundary behind it (peel_first), so the
        // next-tile halo addressing (~60 registers of temporaries) is not in it, and the pending rows (34 registers) are in no other copy.
        auto block = [&](const int cb, auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;
            const bool last_cb = FIRST ? false : cb + 1 == c_end;
            const bool more = !last_cb || has_next;  // a halo follows this one in the stream
            auto step = [&](auto tap_c) {
                constexpr int tap = decltype(tap_c)::value;
                const int bnext = (bcur + 1) & 3;
                constexpr int ntap = tap == 8 ? 0 : tap + 1;
                const int nast = tap == 8 ? hs ^ 1 : hs;
                // The weight / halo streams run straight across tile boundaries: the next tile's first halo and
                // first three weight tiles are fetched under the current tile's last k-steps, so a workgroup pays the
                // global-memory latency of a prologue once per launch, not once per tile.  Past the very end the
                // weight DMA re-fetches the last tile into a free stage and the fragment prefetch reads
                // stale-but-valid LDS; neither result is used (straight-line body, no data-dependent branches).
                auto issue_dma = [&]() {
                    if constexpr (tap == 4) {
                        if (kt == 4) {                                                                 // workgroup-uniform
                            if constexpr (SPLIT) nxt2 = tl + 2 * gx < chunk_len ? decode(tl + 2 * gx) : nxt;
                            else nxt2 = tl + 2 * gx < chunk_len ? advance(nxt) : nxt;
                        }
                    }
                    const int k3 = kt + wdist;
                    const bool wrap = k3 >= nkt;
                    if (!(EARLY_W && tap == 0 && post_ep == 2))      // (already issued by the epilogue)
                        issue_B(wrap ? nxt.n0 : cur.n0, wrap && has_next ? t_cb0(nxt) : t_cb0(cur), wrap ? (has_next ? k3 - nkt : nkt - 1) : k3, (bcur + wdist) & 3);
                    // the next halo goes out two instructions per step over taps 0-2, AFTER the step's weights (needed first)
                    if constexpr (tap <= 2) {
                        if (more) {
                            if constexpr (tap == 0) {
                                // The tile coordinates go through an empty asm INSIDE the branch: the address arithmetic (~250 VALU with
                                // 64-bit multiplies) is invariant in the channel-block loop, and the compiler hoisted it - both variants,
                                // speculatively - to the top of every tile, between the epilogue and the first MFMA (tools/halo_stamps.py:
                                // ~2.6 k cycles there); here it runs between two MFMA clusters of a step.  A next list entry on the same
                                // patch (the other N-tile: N-tile fastest) keeps the addresses it has.
                                if constexpr (!FIRST) {
                                if (last_cb) {
                                    if (p.C1 > 0 || nxt.tb != cur.tb || nxt.y0 != cur.y0 || nxt.x0 != cur.x0) {
                                        TileH tn = nxt;
                                        asm volatile("" : "+s"(tn.tb), "+s"(tn.y0), "+s"(tn.x0));
                                        // (un-split tiles end in segment 1 of a concatenated input and start in segment 0)
                                        if (!SPLIT && !has_gn && (p.C1 == 0 || p.C1 == p.C0) && same_borders(cur, tn))      // (has_gn: the coefficient rows move with the image too)
                                            halo_shift(tile_distance(cur, tn, p.C0) + (p.C1 > 0 ? reinterpret_cast<const char*>(p.x0) - reinterpret_cast<const char*>(p.x1) : (int64_t)0));
                                        else
                                            halo_addr(tn, t_seg(nxt));
                                    }
                                } else if (p.C1 > 0 && cb + 1 == cbs1) {
                                    if (p.C1 == p.C0 && !has_gn) {
                                        halo_shift(reinterpret_cast<const char*>(p.x1) - reinterpret_cast<const char*>(p.x0));
                                    } else {
                                        TileH tn = cur;
                                        asm volatile("" : "+s"(tn.tb), "+s"(tn.y0), "+s"(tn.x0));
                                        halo_addr(tn, 1);
                                    }
                                }
                                }
                                if (coef_wave) issue_coef(last_cb ? 0 : cb + 1, hs ^ 1);     // BEFORE the halo rows: retired first
                            }
                            issue_A(last_cb ? t_cb0(nxt) : cb + 1, hs ^ 1, std::integral_constant<int, 2 * tap>{}, std::integral_constant<int, 2>{});
                        }
                    }
                };
                // GroupNorm prologue: halo instructions (2t, 2t+1) of the NEXT channel block were issued at tap t and retired by this
                // wave's wait at tap t + 2 (the coefficient piece: tap 0 -> retired at tap 1, published by that step's barrier).
                // They are normalised in place one per step at taps 3..7 (instruction j = tap - 3; the sixth, which only wave 0 has,
                // also at tap 7): the data read is issued here, ahead of the first MFMA cluster, the arithmetic follows the DMA issue.
                // The barriers of those steps publish the result before tap 8 prefetches the next block's first fragments.
                constexpr bool xf_tap = has_xf && tap >= 3 && tap <= 7;
                const bool xf_on = xf_tap && more;
                uint4 xd0 = uint4{0, 0, 0, 0}, xd1 = uint4{0, 0, 0, 0};
                if constexpr (xf_tap) {
                    if (xf_on) {
                        xd0 = xload(hs ^ 1, tap - 3);
                        if constexpr (tap == 7) { if (wave == 0) xd1 = xload(hs ^ 1, 5); }
                    }
                }
                // Each half: 8 fragment reads for a LATER cluster + 16 MFMAs.  The sched_group_barrier pattern makes the
                // backend interleave them as [1 ds_read, 2 MFMA] x 8 instead of "all reads, then all MFMAs": an MFMA holds
                // the SIMD's issue port for 8 of its 16 cycles, so a read slipped into each gap costs ~nothing and a wave
                // keeps the matrix pipe fed on its own (stamps: the read bursts used to cost 100-290 cycles per half with
                // the pipe idle).
                auto xf_mid = [&]() {
                    if constexpr (xf_tap) {
                        if (xf_on) {
                            // hard scheduling fences: the block's LDS reads must not be drawn into the [1 ds_read, 2 MFMA] groups around
                            // it (that interleave kept 60+ registers live and spilled INSIDE the k-loop; a spill reload is a vector-
                            // memory load whose wait drains every DMA in flight)
                            __builtin_amdgcn_sched_barrier(0);
                            xfinish(hs ^ 1, tap - 3, xd0);
                            if constexpr (tap == 7) { if (wave == 0) xfinish(hs ^ 1, 5, xd1); }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                };
                if constexpr (has_x3) {
                    // (fa0, fb0) = this step's hi fragments, loaded during the previous step.  First cluster: hi*hi under the reads of
                    // the lo fragments; second cluster: hi*lo + lo*hi (32 MFMAs) under the reads of the NEXT step's hi fragments,
                    // which go to a third register set because this step's are still operands.
                    load_frags(fa1, fb1, hs, bcur, tap_c, K1{});
                    mma16(fa0, fb0);
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    }
                    issue_dma();
                    xf_mid();
                    load_frags(fa2, fb2, nast, bnext, std::integral_constant<int, ntap>{}, K0{});
                    mma16(fa0, fb1);                              // hi(input) x lo(weights)
                    mma16(fa1, fb0);                              // lo(input) x hi(weights)
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) { fa0[q] = fa2[q]; fb0[q] = fb2[q]; }
                } else {
                load_frags(fa1, fb1, hs, bcur, tap_c, K1{});      // this step's second half
                mma16(fa0, fb0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
                // DMA issue (~90 cycles per instruction for the issuing wave) sits BETWEEN the two MFMA clusters: a step
                // starts with matrix work that is already in registers, and the first-dispatched waves 0-3 (which run
                // ahead of their SIMD partners by ~150 cycles after every barrier) issue while waves 4-7 still compute.
                issue_dma();
                if constexpr (FIRST && tap < 8) {
                    if (pend) *reinterpret_cast<uint4*>(pend_op + (tap >> 1) * ((int64_t)p.Wout * p.Cout) + (tap & 1) * 8) = pend_ref(tap_c);   // row tap of the previous tile
                }
                xf_mid();
                load_frags(fa0, fb0, nast, bnext, std::integral_constant<int, ntap>{}, K0{});   // next step's first half
                mma16(fa1, fb1);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
                }
                // retire weights kt+2; instructions younger than them may stay in flight:
                // this step's weights kt+3 (NB) and the halo instructions issued at this step or the previous one
                // (issue order per step: weights, then 2 halo instructions at taps 0-2)
                if (FIRST && pend) {
                    const int allow = (tap == 0 || tap == 3) ? (more ? NB + 2 : NB) : (tap == 1 || tap == 2) ? (more ? NB + 4 : NB) : NB;
                    switch (allow + (tap == 0 ? pend : tap == 8 ? 1 : 2)) {          // (see DEFER above; pend = 1 + atomics)
                        case NB + 1: dma_wait_h<NB + 1>(); break;
                        case NB + 2: dma_wait_h<NB + 2>(); break;
                        case NB + 3: dma_wait_h<NB + 3>(); break;
                        case NB + 4: dma_wait_h<NB + 4>(); break;
                        case NB + 5: dma_wait_h<NB + 5>(); break;
                        default: dma_wait_h<NB + 6>(); break;
                    }
                    if constexpr (tap == 8) pend = 0;
                }
                else if constexpr (tap == 0) {
                    if (EARLY_W && post_ep == 2) {
                        // younger than the weights of step 2 (which this wait is for): the early weight tile, 8 row stores, this step's halo
                        if (more) dma_wait_h<NB + 8 + 2>(); else dma_wait_h<NB + 8>();
                        post_ep = 1;
                    }
                    else if (more) { if (coef_wave) dma_wait_h<NB + 3>(); else dma_wait_h<NB + 2>(); } else dma_wait_h<NB>();
                }
                else if constexpr (tap == 1) {
                    if (EARLY_W && post_ep == 1) {
                        // younger than the early weight tile: 8 row stores, halo of steps 0 and 1, this step's weights
                        if (more) dma_wait_h<8 + NB + 4>(); else dma_wait_h<8 + NB>();
                        post_ep = 0;
                    }
                    else if (more) dma_wait_h<NB + 4>(); else dma_wait_h<NB>();
                }
                else if constexpr (tap == 3) { if (more) dma_wait_h<NB + 2>(); else dma_wait_h<NB>(); }
                else if constexpr (tap == 2) { if (more) dma_wait_h<NB + 4>(); else dma_wait_h<NB>(); }
                else dma_wait_h<NB>();
                __syncthreads();
                bcur = bnext;
                ++kt;
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
            hs ^= 1;
        };
        int cb = t_cb0(cur);
        if constexpr (peel_first) { block(cb, std::true_type{}); ++cb; }
        for (; cb < c_end; ++cb) block(cb, std::false_type{});
        epilogue(cur, nxt, has_next);                // registers -> global, asynchronous stores; no LDS, no barrier
        if (!has_next) break;
        cur = nxt;
        nxt = nxt2;
        tl += gx;
    }
    if constexpr (DEFER) {
        if (pend) {               // the last tile's rows
            const int64_t rstride = (int64_t)p.Wout * p.Cout;
            *reinterpret_cast<uint4*>(pend_op) = pend0;               *reinterpret_cast<uint4*>(pend_op + 8) = pend1;
            *reinterpret_cast<uint4*>(pend_op + rstride) = pend2;     *reinterpret_cast<uint4*>(pend_op + rstride + 8) = pend3;
            *reinterpret_cast<uint4*>(pend_op + 2 * rstride) = pend4; *reinterpret_cast<uint4*>(pend_op + 2 * rstride + 8) = pend5;
            *reinterpret_cast<uint4*>(pend_op + 3 * rstride) = pend6; *reinterpret_cast<uint4*>(pend_op + 3 * rstride + 8) = pend7;
        }
    }
    dma_wait_h<0>();          // the redundant tail fetches
}

template <typename T, bool GN, bool SPLIT = false, bool X3 = false, bool CF = false>
int launch_halo(const KParams& p, hipStream_t stream) {
    static DeviceOnce once;
    const int slot = nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<T, GN, SPLIT, X3, CF>), hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS);
    });
    const int ncu = once.ncu[slot];
    const int nblk = p.B * (p.Hout / PATCH) * (p.Wout / PATCH) * p.NT * (SPLIT ? p.ksplit : 1);
    int grid = nblk < ncu ? nblk : ncu;              // one persistent workgroup per CU (154 KiB of LDS each)
    if (!(p.tuning & (1 << 27)) && nblk > ncu) {
        // The fewest workgroups that finish in the same number of rounds, in whole multiples of 8 (XCDs): 1 600 tiles on 256 CUs are 7
        // rounds whether 256 workgroups walk them (64 of them seven tiles, 192 six) or 232 (seven each, the last round all but full),
        // and on a chip whose clock is set by power the CUs left idle for the whole launch buy the others a higher one - cfg 3's
        // launches at batch 200 (1 600 / 400 tiles) 1.5-2.9 % faster, whole-round launches unchanged (tuning bit 27 = one per CU: A/B,
        // profiles/r05_summary.md section 8)
        const int rounds = (nblk + ncu - 1) / ncu;
        const int g = ((nblk + rounds - 1) / rounds + 7) & ~7;
        if (g < grid) grid = g;
    }
    hipLaunchKernelGGL((conv_halo_kernel<T, GN, SPLIT, X3, CF>), dim3(grid), dim3(HT), HALO_LDS, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { nlc_set_error("nlc_conv2d(halo): launch failed: %s", hipGetErrorString(e)); return NLC_ELAUNCH; }
    return NLC_OK;
}

}  // namespace


static bool halo_eligible(const KParams& p, int dtype, bool* forced_out) {
    // per-call policy (nlc_conv_desc.policy): no process-wide switch, so the dispatch a test forces and the dispatch a
    // benchmark measures are chosen by the caller, launch by launch
    const bool forced = p.policy == NLC_CONV_FORCE_HALO;
    if (forced_out) *forced_out = forced;
    if (p.policy == NLC_CONV_NO_HALO || p.policy == NLC_CONV_GENERIC) return false;
    if (!(p.KH == 3 && p.KW == 3 && p.pad_t == 1 && p.pad_l == 1 && p.stride == 1)) return false;
    const int HL = p.ups ? 2 * p.Hin : p.Hin, WL = p.ups ? 2 * p.Win : p.Win;
    if (p.Hout % PATCH || p.Wout % PATCH || p.Hout != HL || p.Wout != WL) return false;
    const int kbe = nlc_is16(dtype) ? MmaH<bf16_raw>::KBE : MmaH<float>::KBE;
    if (p.C0 % kbe || p.C1 % kbe || p.Cin_pad / kbe > 128) return false;
    if ((int64_t)p.B * p.Hout * p.Wout >= (1ll << 31)) return false;
    return true;
}

static int halo_tiles(const KParams& p) { return p.B * (p.Hout / PATCH) * (p.Wout / PATCH) * p.NT; }

// Split-K (SPLIT instantiation): launches with fewer tiles than CUs and at least four 64-channel blocks per
// split - the 16x16 level of ADM-256 at B = 16 (1024->1024: 128 tiles x 2, 1024->512: 64 tiles x 4).  bf16, whole N-tiles, no
// GroupNorm prologue; the caller opts in by passing the workspace (nlc_conv2d_workspace_bytes).  tuning bit 7 disables it (A/B).
int nlc_conv_halo_ksplit(const KParams& p, int dtype) {
    if (!nlc_is16(dtype) || p.gn_coef || (p.Cout % BN) != 0 || (p.tuning & (128 | 2048))) return 1;       // bit 11: no split-K anywhere (A/B, stress test)
    if (!halo_eligible(p, dtype, nullptr)) return 1;
    const int tiles = halo_tiles(p), ncb = p.Cin_pad / MmaH<bf16_raw>::KBE;
    if (tiles >= 256) return 1;
    int ks = 256 / tiles;
    if (ks > 4) ks = 4;
    if (ks > ncb / 4) ks = ncb / 4;                  // >= 4 channel blocks (36 k-steps) per split: 256->1024 @16^2 at 2 per split lost 12 %
    if (ks < 2 || tiles * ks < 128) return 1;
    return ks;
}

// the un-split kernel pays from 128 tiles on: at 128-255 (the 16x16 level of ADM-256 at B = 16) half the CUs idle, and it still
// beats conv_fast + split-K + reduce (1024->1024: 100 vs 114 us, 512->1024: 55 vs 70, 256->1024: 34 vs 52; 2048->1024 level)
static bool halo_plain_ok(const KParams& p, int dtype) {
    bool forced = false;
    if (!halo_eligible(p, dtype, &forced)) return false;
    return forced || halo_tiles(p) >= 128;
}

int nlc_conv_halo_plain_ok(const KParams& p, int dtype) { return halo_plain_ok(p, dtype) ? 1 : 0; }

int nlc_conv_halo_prologue_ok(const KParams& p, int dtype) {
    return dtype == NLC_BF16 && halo_plain_ok(p, dtype) ? 1 : 0;
}

// GroupNorm statistics ride along when the halo kernel runs in 16 bits with NHWC output and whole 128-channel N-tiles (split-K
// launches too: from the epilogue of the workgroup that arrives last at the tile)
int nlc_conv_halo_stats_partials(const KParams& p, int dtype) {
    if (!nlc_is16(dtype) || p.out_mode != NLC_OUT_NHWC || (p.Cout % BN) != 0) return 0;
    if (nlc_conv_halo_ksplit(p, dtype) <= 1 && !halo_plain_ok(p, dtype)) return 0;
    return 1;
}

// 3x3 / stride 1 / pad 1 (optionally on the nearest-2x upsampled input), output H and W multiples of 16, enough tiles to fill the chip.
// nlc_conv_desc.policy: NLC_CONV_NO_HALO disables, NLC_CONV_FORCE_HALO forces (for eligible shapes) regardless of the tile count.
// p.ksplit > 1 (set by nlc_conv2d when nlc_conv_halo_ksplit asks for it and the caller passed the workspace): split-K + reduce.
int nlc_conv_halo_dispatch(const KParams& p, int dtype, hipStream_t stream) {
    if (p.ksplit > 1) {
        if (nlc_conv_halo_ksplit(p, dtype) != p.ksplit || !p.partial) return NLC_EUNSUPPORTED;
        if (dtype == NLC_F16) return launch_halo<f16_raw, false, true>(p, stream);
        return launch_halo<bf16_raw, false, true>(p, stream);
    }
    if (!halo_plain_ok(p, dtype)) return NLC_EUNSUPPORTED;
    // CF instantiation: whole N-tiles, bias (and embedding) readable as aligned float4
    const int kbe = MmaH<bf16_raw>::KBE;
    const bool cf = nlc_is16(dtype) && !p.gn_coef && (p.Cout % BN) == 0 && p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0 &&
                    p.out_mode == NLC_OUT_NHWC && p.act == NLC_ACT_NONE && p.Cin_pad / kbe >= 2 && !(p.C1 > 0 && p.C0 / kbe == 1) &&
                    (!p.emb || ((reinterpret_cast<uintptr_t>(p.emb) & 15) == 0 && (p.emb_stride & 3) == 0));
    if (dtype == NLC_BF16) return p.gn_coef ? launch_halo<bf16_raw, true>(p, stream) : cf ? launch_halo<bf16_raw, false, false, false, true>(p, stream) : launch_halo<bf16_raw, false>(p, stream);
    if (dtype == NLC_F16) return cf ? launch_halo<f16_raw, false, false, false, true>(p, stream) : launch_halo<f16_raw, false>(p, stream);      // (GroupNorm prologue: bf16 instantiation only - it is off by default)
    if (p.math == NLC_MATH_F16X3) return launch_halo<float, false, false, true>(p, stream);
    return launch_halo<float, false>(p, stream);
}
