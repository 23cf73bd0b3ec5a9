// Offsets table handed to the chain kernels by value (kernarg segment -> scalar loads), so that layer
// loops can index segments / biases at run time without a per-TU __constant__ copy.
#pragma once
#include "arch.h"

namespace es {

struct Tabs {
    int boff[NETS * LAYERS];       // bias offset inside the effective-weight buffer
    int woff[NETS * LAYERS];       // weight offset inside the effective-weight buffer
    unsigned segoff[SEG_COUNT];    // float4 offset of each packed segment
    unsigned p16off[P16_COUNT];    // float4 offset of each 16x16x4-packed query segment
};

inline Tabs make_tabs() {
    Tabs t;
    int w = 0;
    for (int n = 0; n < NETS; ++n)
        for (int l = 0; l < LAYERS; ++l) {
            t.woff[n * LAYERS + l] = w;
            t.boff[n * LAYERS + l] = w + LAYER_N[n][l] * LAYER_K[n][l];
            w += LAYER_N[n][l] * (1 + LAYER_K[n][l]);
        }
    size_t off = 0;
    for (int i = 0; i < SEG_COUNT; ++i) {
        t.segoff[i] = (unsigned)off;
        off += (size_t)seg_kg(SEGS[i]) * seg_nt(SEGS[i]) * 64;
    }
    for (int i = 0; i < P16_COUNT; ++i) t.p16off[i] = (unsigned)p16_off4(i);
    return t;
}

}  // namespace es
