// EndoSurf network architecture tables, flat-parameter layout and packed-weight segment table.
//
// Single source of truth for every offset used by the HIP kernels, the C ABI and (through the
// es_layout_* queries) the Python host.  The architecture is the one every EndoSurf config of
// the reference ships (configs/endosurf/baseline/base_pull.yml:40-82): 9-layer, 256-wide MLPs
// with one skip at layer 4, frequency encodings L = 6/6/10/4.
#pragma once
#include <cstddef>
#include <cstdint>

namespace es {

constexpr int NETS = 3;      // 0 = deform_network, 1 = sdf_network, 2 = color_network
constexpr int LAYERS = 9;
constexpr int HID = 256;
constexpr int NET_D = 0, NET_S = 1, NET_C = 2;

// in-dim (K) / out-dim (N) per layer, reference nn.Linear shapes weight_v[N][K]
// (build_mlp_idr utils.py:63-111 for deform; build_mlp_nerf utils.py:11-60 for sdf/colour).
constexpr int LAYER_K[NETS][LAYERS] = {
    {52, 256, 256, 256, 256, 256, 256, 256, 256},
    {39, 256, 256, 256, 295, 256, 256, 256, 256},
    {349, 256, 256, 256, 605, 256, 256, 256, 256},
};
constexpr int LAYER_N[NETS][LAYERS] = {
    {256, 256, 256, 204, 256, 256, 256, 256, 3},
    {256, 256, 256, 256, 256, 256, 256, 256, 257},
    {256, 256, 256, 256, 256, 256, 256, 256, 3},
};

// ---- flat parameter buffer: for net, for layer: bias[N], weight_g[N], weight_v[N*K]; then variance ----
constexpr int layer_param_floats(int net, int l) { return LAYER_N[net][l] * (2 + LAYER_K[net][l]); }
constexpr int param_layer_off(int net, int l) {
    int off = 0;
    for (int n = 0; n < NETS; ++n)
        for (int k = 0; k < LAYERS; ++k) {
            if (n == net && k == l) return off;
            off += layer_param_floats(n, k);
        }
    return off;
}
constexpr int PARAM_VARIANCE_OFF = param_layer_off(NETS - 1, LAYERS - 1) + layer_param_floats(NETS - 1, LAYERS - 1);
constexpr int PARAM_FLOATS = PARAM_VARIANCE_OFF + 1;   // 1 654 951
static_assert(PARAM_FLOATS == 1654951, "parameter count must match the reference (SURVEY A.2)");
constexpr int param_bias_off(int net, int l) { return param_layer_off(net, l); }
constexpr int param_g_off(int net, int l) { return param_layer_off(net, l) + LAYER_N[net][l]; }
constexpr int param_v_off(int net, int l) { return param_layer_off(net, l) + 2 * LAYER_N[net][l]; }

// ---- effective-weight buffer (and its gradient): for net, for layer: W[N*K] (row-major), b[N] ----
constexpr int weff_layer_off(int net, int l) {
    int off = 0;
    for (int n = 0; n < NETS; ++n)
        for (int k = 0; k < LAYERS; ++k) {
            if (n == net && k == l) return off;
            off += LAYER_N[n][k] * (LAYER_K[n][k] + 1);
        }
    return off;
}
constexpr int weff_w_off(int net, int l) { return weff_layer_off(net, l); }
constexpr int weff_b_off(int net, int l) { return weff_layer_off(net, l) + LAYER_N[net][l] * LAYER_K[net][l]; }
constexpr int WEFF_FLOATS = weff_layer_off(NETS - 1, LAYERS - 1) + LAYER_N[NETS - 1][LAYERS - 1] * (LAYER_K[NETS - 1][LAYERS - 1] + 1);
// The chain kernels request a layer's bias as one value per output COLUMN OF THE 256-WIDE TILE (query.hip bias2 and friends): for the
// one layer with fewer outputs (deform layer 3: 204) the lanes of columns 204..255 read past its bias vector into the next layer's
// weights.  The values are never used (the skip copy overwrites those columns); the reads must stay inside the buffer.
static_assert(weff_b_off(NET_D, 3) + HID <= WEFF_FLOATS, "bias reads of a 256-wide tile must stay inside the effective-weight buffer");

// ---- packed MFMA B-operand segments --------------------------------------------------------------
// One segment = one GEMM operand B[k][n] (k = contraction index, n = output column) stored as
// float4 tiles [nt][g][lane]: element j of the float4 of lane l is B[8g + 2j + (l>>5)][32nt + (l&31)],
// i.e. exactly the B fragment of four consecutive v_mfma_f32_32x32x2_f32 k-steps.
//   dir F (forward):  B[k][n] = scale * W[row0 + n][col0 + k]      (y = x W^T)
//   dir R (reverse):  B[k][n] = scale * W[row0 + k][col0 + n]      (x_adj = y_adj W)
struct SegDesc {
    int net, layer, dir;     // dir: 0 = F, 1 = R
    int row0, col0;          // window of the [N][K] effective-weight matrix
    int kreal, nreal;        // valid extent along GEMM-k / GEMM-n (rest is zero padding)
    int skip_scale;          // 1 -> multiply by 1/sqrt(2) (skip layer input scaling folded into the weights)
};
constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int seg_kg(const SegDesc& s) { return cdiv(s.kreal, 8); }
constexpr int seg_nt(const SegDesc& s) { return cdiv(s.nreal, 32); }

enum Seg : int {
    // deform forward
    DF0, DF1, DF2, DF3, DF4, DF5, DF6, DF7,
    // deform reverse (input adjoint = output adjoint x W)
    DR0, DR1, DR2, DR3, DR4, DR5, DR6, DR7,
    // sdf forward
    SF0, SF1, SF2, SF3, SF4M, SF4A, SF5, SF6, SF7, SF8F,
    // sdf reverse
    SR0, SR1, SR2, SR3, SR4M, SR4A, SR5, SR6, SR7, SR8F,
    // colour forward
    CF0S, CF0F, CF1, CF2, CF3, CF4H, CF4S, CF4F, CF5, CF6, CF7,
    // colour reverse
    CR0S, CR0F, CR1, CR2, CR3, CR4H, CR4S, CR4F, CR5, CR6, CR7,
    SEG_COUNT
};

constexpr SegDesc SEGS[SEG_COUNT] = {
    // net l dir row0 col0 kreal nreal skip
    {0, 0, 0, 0, 0, 52, 256, 0},   // DF0
    {0, 1, 0, 0, 0, 256, 256, 0},  // DF1
    {0, 2, 0, 0, 0, 256, 256, 0},  // DF2
    {0, 3, 0, 0, 0, 256, 204, 0},  // DF3 (204 outputs, columns 204..255 are zero)
    {0, 4, 0, 0, 0, 256, 256, 1},  // DF4 (input = [h(204) | enc(52)] / sqrt2)
    {0, 5, 0, 0, 0, 256, 256, 0},  // DF5
    {0, 6, 0, 0, 0, 256, 256, 0},  // DF6
    {0, 7, 0, 0, 0, 256, 256, 0},  // DF7
    {0, 0, 1, 0, 0, 256, 52, 0},   // DR0 (adjoint of the 52-wide encoding input: the VJP sweep J^T g)
    {0, 1, 1, 0, 0, 256, 256, 0},  // DR1
    {0, 2, 1, 0, 0, 256, 256, 0},  // DR2
    {0, 3, 1, 0, 0, 204, 256, 0},  // DR3 (contract over the 204 outputs)
    {0, 4, 1, 0, 0, 256, 256, 1},  // DR4
    {0, 5, 1, 0, 0, 256, 256, 0},  // DR5
    {0, 6, 1, 0, 0, 256, 256, 0},  // DR6
    {0, 7, 1, 0, 0, 256, 256, 0},  // DR7
    {1, 0, 0, 0, 0, 39, 256, 0},   // SF0
    {1, 1, 0, 0, 0, 256, 256, 0},  // SF1
    {1, 2, 0, 0, 0, 256, 256, 0},  // SF2
    {1, 3, 0, 0, 0, 256, 256, 0},  // SF3
    {1, 4, 0, 0, 0, 256, 256, 1},  // SF4M (hidden part of the skip input)
    {1, 4, 0, 0, 256, 39, 256, 1}, // SF4A (encoding part of the skip input)
    {1, 5, 0, 0, 0, 256, 256, 0},  // SF5
    {1, 6, 0, 0, 0, 256, 256, 0},  // SF6
    {1, 7, 0, 0, 0, 256, 256, 0},  // SF7
    {1, 8, 0, 1, 0, 256, 256, 0},  // SF8F (feature rows 1..256 of the 257-row last layer)
    {1, 0, 1, 0, 0, 256, 39, 0},   // SR0
    {1, 1, 1, 0, 0, 256, 256, 0},  // SR1
    {1, 2, 1, 0, 0, 256, 256, 0},  // SR2
    {1, 3, 1, 0, 0, 256, 256, 0},  // SR3
    {1, 4, 1, 0, 0, 256, 256, 1},  // SR4M
    {1, 4, 1, 0, 256, 256, 39, 1}, // SR4A
    {1, 5, 1, 0, 0, 256, 256, 0},  // SR5
    {1, 6, 1, 0, 0, 256, 256, 0},  // SR6
    {1, 7, 1, 0, 0, 256, 256, 0},  // SR7
    {1, 8, 1, 1, 0, 256, 256, 0},  // SR8F
    {2, 0, 0, 0, 0, 93, 256, 0},   // CF0S (enc10(x_c) 63 | g_c 3 | enc4(d_c) 27)
    {2, 0, 0, 0, 93, 256, 256, 0}, // CF0F (geometry feature 256)
    {2, 1, 0, 0, 0, 256, 256, 0},  // CF1
    {2, 2, 0, 0, 0, 256, 256, 0},  // CF2
    {2, 3, 0, 0, 0, 256, 256, 0},  // CF3
    {2, 4, 0, 0, 0, 256, 256, 1},  // CF4H
    {2, 4, 0, 0, 256, 93, 256, 1}, // CF4S
    {2, 4, 0, 0, 349, 256, 256, 1},// CF4F
    {2, 5, 0, 0, 0, 256, 256, 0},  // CF5
    {2, 6, 0, 0, 0, 256, 256, 0},  // CF6
    {2, 7, 0, 0, 0, 256, 256, 0},  // CF7
    {2, 0, 1, 0, 0, 256, 93, 0},   // CR0S
    {2, 0, 1, 0, 93, 256, 256, 0}, // CR0F
    {2, 1, 1, 0, 0, 256, 256, 0},  // CR1
    {2, 2, 1, 0, 0, 256, 256, 0},  // CR2
    {2, 3, 1, 0, 0, 256, 256, 0},  // CR3
    {2, 4, 1, 0, 0, 256, 256, 1},  // CR4H
    {2, 4, 1, 0, 256, 256, 93, 1}, // CR4S
    {2, 4, 1, 0, 349, 256, 256, 1},// CR4F
    {2, 5, 1, 0, 0, 256, 256, 0},  // CR5
    {2, 6, 1, 0, 0, 256, 256, 0},  // CR6
    {2, 7, 1, 0, 0, 256, 256, 0},  // CR7
};

// offset of a segment inside the packed buffer, in float4 units
constexpr size_t seg_off4(int id) {
    size_t off = 0;
    for (int i = 0; i < id; ++i) off += (size_t)seg_kg(SEGS[i]) * seg_nt(SEGS[i]) * 64;
    return off;
}
constexpr size_t PACKED_FLOAT4 = seg_off4(SEG_COUNT);
constexpr size_t PACKED_FLOATS = PACKED_FLOAT4 * 4;

// ---- second packing of the forward query segments for v_mfma_f32_16x16x4_f32 (16-point tiles of the latency-bound
// small-batch SDF query): float4 tiles [nt16][g][lane], element j of lane l = B[16g + 4j + (l>>4)][16 nt16 + (l&15)].
constexpr int P16_SEGS[] = {DF0, DF1, DF2, DF3, DF4, DF5, DF6, DF7, SF0, SF1, SF2, SF3, SF4M, SF4A, SF5, SF6, SF7};
constexpr int P16_COUNT = sizeof(P16_SEGS) / sizeof(int);
constexpr int p16_kg(int i) { return cdiv(SEGS[P16_SEGS[i]].kreal, 16); }
constexpr size_t p16_off4(int i) {     // float4 offset inside the packed buffer (after the 32x32 segments)
    size_t off = PACKED_FLOAT4;
    for (int k = 0; k < i; ++k) off += (size_t)p16_kg(k) * 16 * 64;
    return off;
}
constexpr size_t PACKED_TOTAL_FLOAT4 = p16_off4(P16_COUNT);
constexpr size_t PACKED_TOTAL_FLOATS = PACKED_TOTAL_FLOAT4 * 4;

constexpr float INV_SQRT2 = 0.70710678118654752440f;

}  // namespace es
