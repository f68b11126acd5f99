// Parameter blocks of the small-map 3x3 kernel (conv_small.hip): shared by the nlc_conv2d dispatch (conv_igemm.hip) and the kernel.
#pragma once
#include "conv_params.h"

// GroupNorm (+FiLM) (+SiLU) of the convolution's INPUT from the ride-along totals of the launches that produced it
// (nlc_conv_desc.gn_in); tot0 == nullptr: the input is used as it is.  Field names follow GNParams (groupnorm.hip) so that
// gn_group_from_totals_t serves both.
struct GnIn {
    const long long* tot0; const long long* tot1;
    int tsh0, tsh1;             // log2 of the totals' granules (2 | 3)
    int C0, C1, gs;
    double invN;                // 1 / (H * W * gs)
    float eps;
    const float* gamma; const float* beta; const float* scale; const float* shift; int ss_stride;
    int act;                    // NLC_ACT_NONE | NLC_ACT_SILU
    FastDiv div_gs;
};

struct SmallGeom {
    int wm;                     // wave rows of a workgroup: 2 = 128-pixel tiles (256 threads), 4 = 256-pixel tiles (512 threads)
    int MT;                     // pixel tiles (M / (64 wm))
    int nb, ks;                 // 64-channel blocks per workgroup (k-slice), number of k-slices
    int nwst;                   // weight stages in LDS (3 | 4 | 6 | 8)
    int dist;                   // split-K reduction distributed over the tile's ks workgroups (all resident) instead of by the last arriver
    int TR, SR, nseg;           // tile rows of the stacked (B * H)-row image; rows per image segment; segments (images) per tile
    int slots;                  // nseg * (SR + 2) * (W + 2) halo positions per channel block
    FastDiv div_w, div_wp, div_h, div_sr, div_segslots;
};

struct SmallParams {
    KParams k;
    GnIn gn;
    SmallGeom geo;
    int out_sc1;                // stores of the output are write-through (the two-convolution launch hands h to other CUs)
};

bool nlc_conv_small_geom(const KParams& p, int dtype, SmallGeom& g);
int64_t nlc_conv_small_split_bytes(const KParams& p, const SmallGeom& g);
int nlc_conv_small_dispatch(const SmallParams& sp, int dtype, hipStream_t stream);
// one launch for a ResBlock's two convolutions (sp1's output is sp2's input); bar: two zeroed ints
int nlc_resblock_small_dispatch(const SmallParams& sp1, const SmallParams& sp2, int* bar, int dtype, hipStream_t stream);
