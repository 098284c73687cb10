// Implicit-GEMM convolution / linear layer on the CDNA4 matrix cores (gfx950).
//
// One kernel family serves every contraction of the SD-1.5 hot path: Linear and 1x1 conv
// (taps = 1), the ResBlock / VAE 3x3 convs (taps = 9: nine shifted 1x1 contractions over the
// same pixel tile, zero padding by predicate), stride-2 downsamples (symmetric pad 1 for the
// UNet, openaimodel.py:138-164; pad (0,1,0,1) for the VAE, model.py:73-77), the nearest-2x
// upsample fused into the gather (openaimodel.py:95-123), and the data-gradient of each of
// them (same kernel, transposed/flipped weight pack; zero-insert gather for stride 2).
//
// Layout: activations are pixel-major ("NHWC"): x[(b*H + y)*W + x][c], c contiguous, either
// bf16 or f32 (converted to bf16 on the way into LDS).  Weights are pre-packed bf16
// [tap][Cout][Cin] (Cin contiguous), so both MFMA operands are K-contiguous.
//
// Tile: 128 pixels x BN output channels x 64 input channels per step, 256 threads = 4 waves in
// a 2 (pixel halves) x 2 (channel halves) arrangement, v_mfma_f32_16x16x32_bf16 with the
// WEIGHTS as the A operand and the PIXELS as the B operand: an accumulator register quad then
// holds 4 consecutive output channels of one pixel, i.e. one 16-byte store into the pixel-major
// output.  LDS tiles are [row][64 bf16] with 128-byte rows, 16-byte chunk index XOR (row & 7)
// so that ds_read_b128 fragment reads are bank-conflict free (guide T2), register-staged
// double buffering with one barrier per K-step (guide T14).
//
// Epilogue (fused): * alpha, + bias[c], + chan_add[b][c] (the ResBlock time-embedding add,
// openaimodel.py:271-277), + residual[pixel][c] (skip / transformer residual), written as f32
// and/or bf16.  Split-K (grid.z) for the small-pixel-count / long-K layers of the UNet's low
// resolutions: every split writes its raw fp32 tile to a workspace slab and a second kernel sums the
// slabs in a fixed order and applies the epilogue -- no float atomics, bit-reproducible.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct ConvParams {
    const void* x;          // activations (bf16 or f32), pixel-major
    long ldx;               // elements between consecutive pixels
    const uint16_t* w;      // packed bf16 [taps][Cout][Cin]
    const float* bias;      // [Cout] or null
    const float* chan_add;  // [B][ld_ca] or null
    long ld_ca;
    const float* residual;  // f32 [M][ldr] or null
    long ldr;
    float* y32;             // f32 out or null
    long ldy32;
    uint16_t* y16;          // bf16 out or null
    long ldy16;
    int B, Hin, Win, Cin, Hout, Wout, Cout;
    int KH, KW, stride, pad, up;   // up: 0 none, 1 nearest x2, 2 zero-insert x2
    int ktiles_per_tap;            // ceil(Cin / 64)
    int ktiles_total;              // taps * ktiles_per_tap
    int ksplit;                    // grid.z
    int ntiles_n;                  // ceil(Cout / BN)
    int ntiles_m;                  // ceil(M / 128)
    float alpha;
    float* ws;              // split-K slabs [ksplit][M][Cout] f32
    unsigned x_bytes, w_bytes;   // extents of one batch item's activation / weight block (buffer descriptors)
    long batch_stride_x;    // batched mode (B2 > 1 via grid.y): element strides per batch item
    long batch_stride_w;
    long batch_stride_y32;
    long batch_stride_y16;
    // fused GEGLU epilogues of the feed-forward (attention.py:32-59); the 8C pre-activation h is kept in a PERMUTED channel
    // order -- blocks of 16 value channels alternate with the blocks of their 16 gate channels (the weight pack's rows are
    // permuted accordingly) -- so that a lane holds a value quad and its gate quad in two accumulator tiles of the same wave:
    //   epi 1 (ff.net.0.proj forward): y16 <- h (bf16, permuted), z16 <- a * gelu(gate)              [M][Cout / 2]
    //   epi 2 (ff.net.2 data gradient, Cout = 4C): acc = d(a * gelu(gate)); reads h16, y16 <- dh (bf16, permuted) [M][2 Cout]
    int epi;
    uint16_t* z16; long ldz16;
    const uint16_t* h16; long ldh16;
    // GroupNorm statistics of the OUTPUT from the epilogue (adap_conv2d_next_gn_partial; stencil-window kernel, unsplit):
    // gn_part [B][Hout * Wout / 64][32 groups][2] f32 <- (sum, sum of squares) of every group's channels over 64 pixels (one wave's
    // share of a tile) -- the format of the two-pass GroupNorm's statistics pass, which the consumer then skips
    float* gn_part;
    int gn_cpg;             // channels per group (Cout / 32)
    int dbg;                // what-if switches for tuning (env ADAP_CONV_DEBUG; 0 in production): 1 no DMA in the loop,
                            // 2 no MFMA, 4 no epilogue stores -- results are garbage with any of them set
    unsigned long long* clk;  // diagnostic (adap_conv2d_set_clock_probe; NULL in production): per workgroup of the
                              // stencil-window kernel, shader-clock and 100 MHz real-time ticks spent in its K loop
};

#define BM 128
#define BK 64

typedef __attribute__((address_space(3))) void lds_void_t;

// ---------------------------------------------------------------------------------------------
// The fused epilogue shared by every contraction kernel: y = alpha * acc + bias[c] + chan_add[b][c] + residual[m][c], as
// f32 and / or bf16.  acc[i][j] is the accumulator quad of channel tile i and pixel fragment j; `geo` maps (i, j) to the
// output pixel / image / channel of this lane.
// All addend loads of the tile are ISSUED BEFORE the first store.  Written as one load-add-store per quad, the compiler
// must keep program order between a quad's stores and the next quad's loads (y32 may alias residual: the transformer's
// residual adds run in place), so every quad paid a full memory round trip: 0.6 us x 20 quads = 13 us of a 21 us
// launch on the 128 x 160 tile (in-kernel stamps, tools/gemm_timeline.py).  Each lane only ever re-reads addresses it
// writes itself, so hoisting the loads is safe under that aliasing.
// ---------------------------------------------------------------------------------------------
// Phi(g), phi(g) of the exact GELU: erf by Abramowitz-Stegun 7.1.26 on v_rcp_f32 / v_exp_f32 (misc.hip gelu_cdf_pdf: the same
// arithmetic, so the fused and the separate GEGLU kernels agree bit for bit)
__device__ __forceinline__ void conv_gelu_cdf_pdf(float g, float& cdf, float& pdf) {
    const float x = fabsf(g) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
    const float e = __expf(-x * x);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e, g));
    pdf = 0.3989422804014327f * e;
}

__device__ __forceinline__ void unpack_bf16x4(uint2 v, float* f) {
    f[0] = __builtin_bit_cast(float, v.x << 16); f[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, v.y << 16); f[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
}

// Widened bf16 epilogue stores.  A lane's quad of an accumulator tile is 4 channels of one pixel = 8 bytes as bf16, and the four
// lanes that hold a pixel's 16 channels of the tile (lanes l, l + 16, l + 32, l + 48) write 32 contiguous bytes: a wave's store
// instruction covers 16 rows x 32 bytes, and the epilogues of the bf16-output contractions were bound by the NUMBER of such
// instructions, not by bytes (tools/ff_probe.py: FF1's main loop 33 us of an 81 us launch; guide T21).  Two channel-adjacent
// tiles A (tile i) and B (tile i + 1) are regrouped with v_permlane16_swap (odd 16-lane rows of A <-> even rows of B): afterwards
// lane rows 0 / 2 hold quads (0,1) / (2,3) of tile i and rows 1 / 3 those of tile i + 1 -- 16 contiguous bytes per lane, ONE
// dwordx4 store where there were two dwordx2, 16 rows x 64 bytes per instruction.  Same values, same addresses.
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
// the regrouped pair as this lane's 16 bytes; ``c_off``: its channel offset from tile i's first channel (elements)
__device__ __forceinline__ uint4 pair16(uint2 a, uint2 b, int fchunk, int& c_off) {
    swap16(a.x, b.x);
    swap16(a.y, b.y);
    c_off = (fchunk & 1) * 16 + (fchunk >> 1) * 8;
    return make_uint4(a.x, a.y, b.x, b.y);
}
// the inverse for loads: 16 bytes at ``c_off`` -> this lane's quad of tile i (a) and of tile i + 1 (b)
__device__ __forceinline__ void unpair16(uint4 v, uint2& a, uint2& b) {
    uint2 lo = make_uint2(v.x, v.y), hi = make_uint2(v.z, v.w);
    swap16(lo.x, hi.x);
    swap16(lo.y, hi.y);
    a = lo;
    b = hi;
}

// epi 1: accumulator tiles (i, i + 1) of a wave are a block of 16 value channels and the block of their gates
template <int MT, int PT, class Geo>
__device__ __forceinline__ void conv_epilogue_geglu_fwd(const ConvParams& p, const f32x4 (&acc)[MT][PT], const Geo& geo) {
    if constexpr (MT % 2 == 0) {
        // (value block | gate block) of a pair are 64 contiguous bytes of h, the outputs of two pairs 64 contiguous bytes of z:
        // regrouped into 16-byte stores (pair16) when the rows are 16-byte aligned
        const bool wide = (p.ldy16 & 7) == 0 && ((uintptr_t)p.y16 & 15) == 0 && (p.ldz16 & 7) == 0 && ((uintptr_t)p.z16 & 15) == 0 &&
                          (geo.cbase & 31) == 0 && !(p.dbg & 8);
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            int m, b;
            const bool mok = geo.pixel(j, m, b);
            uint2 zz[MT / 2 + 1];
#pragma unroll
            for (int i = 0; i < MT; i += 2) {
                const unsigned ca = (unsigned)geo.chan(i), cg = (unsigned)geo.chan(i + 1);
                const bool cok = (int)cg < p.Cout;                                  // (the same for all lanes of the wave: Cout % 32 == 0)
                float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bg = ba;
                if (p.bias && cok) {
                    ba = *(const float4*)(p.bias + ca);
                    bg = *(const float4*)(p.bias + cg);
                }
                const uint2 ha = make_uint2(pack_bf16x2(acc[i][j][0] * p.alpha + ba.x, acc[i][j][1] * p.alpha + ba.y),
                                            pack_bf16x2(acc[i][j][2] * p.alpha + ba.z, acc[i][j][3] * p.alpha + ba.w));
                const uint2 hg = make_uint2(pack_bf16x2(acc[i + 1][j][0] * p.alpha + bg.x, acc[i + 1][j][1] * p.alpha + bg.y),
                                            pack_bf16x2(acc[i + 1][j][2] * p.alpha + bg.z, acc[i + 1][j][3] * p.alpha + bg.w));
                if (wide && cok) {
                    int c_off;
                    const uint4 w = pair16(ha, hg, geo.fchunk, c_off);
                    if (mok) *(uint4*)(p.y16 + (size_t)m * p.ldy16 + (unsigned)(geo.cbase + i * 16 + c_off)) = w;
                } else if (mok && cok) {
                    *(uint2*)(p.y16 + (size_t)m * p.ldy16 + ca) = ha;
                    *(uint2*)(p.y16 + (size_t)m * p.ldy16 + cg) = hg;
                }
                float a[4], g[4], o[4];
                unpack_bf16x4(ha, a);                  // (from the rounded values: what the backward will read)
                unpack_bf16x4(hg, g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float cdf, pdf;
                    conv_gelu_cdf_pdf(g[e], cdf, pdf);
                    o[e] = a[e] * (g[e] * cdf);
                }
                zz[i / 2] = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
            }
            // z: pair q = (tiles 2q, 2q + 1) writes 16 channels at (cbase / 2 + 16 q); two pairs regroup like two tiles
#pragma unroll
            for (int q = 0; q < MT / 2; q += 2) {
                const unsigned cz0 = (unsigned)(geo.cbase >> 1) + 16u * q;           // first z channel of pair q (this wave)
                if (q + 1 < MT / 2 && wide && geo.cbase + (2 * q + 4) * 16 <= p.Cout) {
                    int c_off;
                    const uint4 w = pair16(zz[q], zz[q + 1], geo.fchunk, c_off);
                    if (mok) *(uint4*)(p.z16 + (size_t)m * p.ldz16 + cz0 + (unsigned)c_off) = w;
                } else {
                    const unsigned ca0 = (unsigned)geo.chan(2 * q), ca1 = (unsigned)geo.chan(2 * q + 2);
                    if (mok && (int)ca0 + 16 < p.Cout)
                        *(uint2*)(p.z16 + (size_t)m * p.ldz16 + (ca0 >> 5) * 16u + (ca0 & 15u)) = zz[q];
                    if (q + 1 < MT / 2) {
                        if (mok && (int)ca1 + 16 < p.Cout)
                            *(uint2*)(p.z16 + (size_t)m * p.ldz16 + (ca1 >> 5) * 16u + (ca1 & 15u)) = zz[q + 1];
                    }
                }
            }
        }
    }
}

// epi 2: acc = d out of the 4C activations; dh = [d out * gelu(gate) | d out * a * gelu'(gate)] in the permuted 8C layout
template <int MT, int PT, class Geo>
__device__ __forceinline__ void conv_epilogue_geglu_bwd(const ConvParams& p, const f32x4 (&acc)[MT][PT], const Geo& geo) {
    // a tile's (value | gate) blocks are 64 contiguous bytes of h and of dh: 16-byte loads / stores regrouped over the four lanes
    // of a pixel (unpair16 / pair16) when the rows are 16-byte aligned
    // (opt-in, ADAP_CONV_DEBUG bit 4: here the regrouped form measured 3-4 % SLOWER -- 58.1 vs 55.7 us at 64 x 64 -- the loads'
    // regrouping sits between the load and the GELU' arithmetic that hides its latency in the narrow form)
    const bool wide = (p.dbg & 16) && (p.ldh16 & 7) == 0 && ((uintptr_t)p.h16 & 15) == 0 && (p.ldy16 & 7) == 0 &&
                      ((uintptr_t)p.y16 & 15) == 0 && (geo.cbase & 15) == 0;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const unsigned c = (unsigned)geo.chan(i);
        const bool cok = (int)c < p.Cout;                                  // (tiles of 16 inside Cout % 16 == 0: wave-uniform)
        const unsigned ch = (c >> 4) * 32u + (c & 15u);                    // the value quad's position in h; its gates: + 16
        const unsigned cht = (unsigned)((geo.cbase + i * 16) >> 4) * 32u;  // ... and the tile's (value | gate) block
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            int m, b;
            const bool mok = geo.pixel(j, m, b);
            uint2 ua = make_uint2(0u, 0u), ug = ua;
            if (wide && cok) {
                const int c_off = (geo.fchunk & 1) * 16 + (geo.fchunk >> 1) * 8;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (mok) v = *(const uint4*)(p.h16 + (size_t)m * p.ldh16 + cht + (unsigned)c_off);
                unpair16(v, ua, ug);
            } else if (mok && cok) {
                ua = *(const uint2*)(p.h16 + (size_t)m * p.ldh16 + ch);
                ug = *(const uint2*)(p.h16 + (size_t)m * p.ldh16 + ch + 16);
            }
            float a[4], g[4], da[4], dg[4];
            unpack_bf16x4(ua, a);
            unpack_bf16x4(ug, g);
            // (the separate kernel read d out as bf16: round it the same way)
            float d[4];
            unpack_bf16x4(make_uint2(pack_bf16x2(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha),
                                     pack_bf16x2(acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha)), d);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float cdf, pdf;
                conv_gelu_cdf_pdf(g[e], cdf, pdf);
                da[e] = d[e] * g[e] * cdf;
                dg[e] = d[e] * a[e] * (cdf + g[e] * pdf);
            }
            const uint2 oa = make_uint2(pack_bf16x2(da[0], da[1]), pack_bf16x2(da[2], da[3]));
            const uint2 og = make_uint2(pack_bf16x2(dg[0], dg[1]), pack_bf16x2(dg[2], dg[3]));
            if (wide && cok) {
                int c_off;
                const uint4 w = pair16(oa, og, geo.fchunk, c_off);
                if (mok) *(uint4*)(p.y16 + (size_t)m * p.ldy16 + cht + (unsigned)c_off) = w;
            } else if (mok && cok) {
                *(uint2*)(p.y16 + (size_t)m * p.ldy16 + ch) = oa;
                *(uint2*)(p.y16 + (size_t)m * p.ldy16 + ch + 16) = og;
            }
        }
    }
}

template <int MT, int PT, class Geo, int JB = (MT * PT > 16 ? PT / 2 : PT)>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const f32x4 (&acc)[MT][PT], const Geo& geo, float* y32,
                                              uint16_t* y16, float (*st)[2] = nullptr) {
    // st (stencil-window kernel with p.gn_part): per channel tile, this lane's sum and sum of squares of the values it writes
    static_assert(PT % JB == 0, "pixel fragments are processed in batches of JB");
    // 32-bit byte offsets against wave-uniform bases (global_load/store saddr form): twenty 64-bit addresses per tensor
    // would otherwise be live across the batch (the host checks that every tensor spans < 4 GiB)
    unsigned cc[MT];
    bool cok[MT];
    float4 bq[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        cc[i] = (unsigned)geo.chan(i);
        cok[i] = (int)cc[i] < p.Cout;
        bq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && cok[i]) bq[i] = *(const float4*)((const char*)p.bias + cc[i] * 4u);
    }
    const bool has_res = p.residual != nullptr, has_ca = p.chan_add != nullptr;
    const unsigned ldr = (unsigned)p.ldr, ldca = (unsigned)p.ld_ca, ld32 = (unsigned)p.ldy32, ld16 = (unsigned)p.ldy16;
    const bool wide16 = y16 && (p.ldy16 & 7) == 0 && ((uintptr_t)y16 & 15) == 0 && (geo.cbase & 7) == 0 && !(p.dbg & 8);   // (wave-uniform)
#pragma unroll
    for (int j0 = 0; j0 < PT; j0 += JB) {
        unsigned mm[JB], bb[JB];
        bool mok[JB];
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            int m, b;
            mok[j] = geo.pixel(j0 + j, m, b);
            mm[j] = (unsigned)m;
            bb[j] = (unsigned)b;
        }
        float4 rq[MT][JB];
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) rq[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_res) {
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    if (mok[j] && cok[i]) rq[i][j] = *(const float4*)((const char*)p.residual + (mm[j] * ldr + cc[i]) * 4u);
        }
        if (has_ca) {
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    if (mok[j] && cok[i]) {
                        const float4 t = *(const float4*)((const char*)p.chan_add + (bb[j] * ldca + cc[i]) * 4u);
                        rq[i][j].x += t.x; rq[i][j].y += t.y; rq[i][j].z += t.z; rq[i][j].w += t.w;
                    }
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            uint2 o16[MT + 1];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                o16[i] = make_uint2(0u, 0u);
                if (!(mok[j] && cok[i])) continue;
                const f32x4 a = acc[i][j0 + j];
                const float v0 = a[0] * p.alpha + bq[i].x + rq[i][j].x, v1 = a[1] * p.alpha + bq[i].y + rq[i][j].y;
                const float v2 = a[2] * p.alpha + bq[i].z + rq[i][j].z, v3 = a[3] * p.alpha + bq[i].w + rq[i][j].w;
                if (st) {
                    // BEFORE the stores, and pinned there: placed after them the compiler builds these sums in the registers the
                    // store still reads (they are dead to it) a handful of instructions later, and gfx950 does not interlock a
                    // VALU write against a global store's data read that late -- one dword of a 16-lane row of the output came
                    // out wrong now and then (tests/test_kernels_gpu.py's bit-reproducibility loop found it)
                    st[i][0] += (v0 + v1) + (v2 + v3);
                    st[i][1] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                    asm volatile("" : "+v"(st[i][0]), "+v"(st[i][1]) :: "memory");
                }
                if (y32) *(float4*)((char*)y32 + (mm[j] * ld32 + cc[i]) * 4u) = make_float4(v0, v1, v2, v3);
                if (y16) {
                    o16[i].x = pack_bf16x2(v0, v1);
                    o16[i].y = pack_bf16x2(v2, v3);
                }
            }
            if (y16) {
                // pairs of channel-adjacent tiles that lie wholly inside Cout go out as one 16-byte store per lane (pair16);
                // (all lanes take part in the regrouping: the predicate is on the store)
#pragma unroll
                for (int i = 0; i < MT; i += 2) {
                    if (i + 1 < MT && wide16 && geo.cbase + (i + 2) * 16 <= p.Cout) {
                        int c_off;
                        const uint4 w = pair16(o16[i], o16[i + 1], geo.fchunk, c_off);
                        if (mok[j]) *(uint4*)((char*)y16 + (mm[j] * ld16 + (unsigned)(geo.cbase + i * 16 + c_off)) * 2u) = w;
                    } else {
                        if (mok[j] && cok[i]) *(uint2*)((char*)y16 + (mm[j] * ld16 + cc[i]) * 2u) = o16[i];
                        if (i + 1 < MT) {
                            if (mok[j] && cok[i + 1]) *(uint2*)((char*)y16 + (mm[j] * ld16 + cc[i + 1]) * 2u) = o16[i + 1];
                        }
                    }
                }
            }
        }
    }
}

// split-K: this split's raw accumulators to its slab (no epilogue; splitk_reduce_kernel sums the slabs in a fixed order)
template <int MT, int PT, class Geo>
__device__ __forceinline__ void conv_store_slab(const ConvParams& p, const f32x4 (&acc)[MT][PT], const Geo& geo, float* slab) {
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        int m, b;
        if (!geo.pixel(j, m, b)) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int c0 = geo.chan(i);
            if (c0 < p.Cout)
                *(float4*)(slab + (size_t)m * p.Cout + c0) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

// GroupNorm statistics of a wave's share of the output tile (p.gn_part; the consumer's statistics pass -- 4 or 2 bytes per element
// of HBM reads -- is then skipped).  Every contraction kernel has the same wave layout: a wave owns 64 pixels x MT channel tiles
// of 16; a lane's quad = 4 consecutive channels of one pixel, the 16 lanes of a row = the 16 pixels of a fragment.  Wave-local
// and in a fixed order: lane sums over the wave's pixel fragments (conv_epilogue's st) -> butterfly over the 16 pixels -> over
// the 1 / 2 / 4 quads of a group (4 / 8 / 16 channels per group) -> one record per (wave's pixel range, group).  No LDS, no
// barrier: `chunk` numbers the wave's pixel range within the whole output (image-major).
template <int MT>
__device__ __forceinline__ void epilogue_gn_stats(const ConvParams& p, float (&st)[MT][2], int frow, int fchunk, int wave_c0,
                                                  size_t chunk) {
    const int qpg = p.gn_cpg >> 2;              // quads per group: 1, 2 or 4
    // The butterfly's ds_bpermute results land in registers the register allocator has just freed -- the data registers of
    // the epilogue's last global stores.  gfx950 does not interlock an LDS return against a VMEM store still reading its data:
    // without this wait one dword of a 16-lane row of the OUTPUT came out wrong now and then (found by the bit-reproducibility
    // check of tests/test_kernels_gpu.py; the same family as the store hazard noted in norms.hip).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float v = st[i][k];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
            if (qpg >= 2) v += __shfl_xor(v, 16, 64);
            if (qpg == 4) v += __shfl_xor(v, 32, 64);
            st[i][k] = v;
        }
    if (frow == 0 && (fchunk & (qpg - 1)) == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int g = (wave_c0 + i * 16 + fchunk * 4) / p.gn_cpg;
            *(float2*)(p.gn_part + (chunk * 32 + g) * 2) = make_float2(st[i][0], st[i][1]);
        }
    }
}

// lane -> (pixel, image, channel) of the plain implicit-GEMM tiling: pixel fragments of 16 consecutive output pixels
struct GemmGeo {
    int mbase, cbase, M, HWo, frow, fchunk;
    __device__ __forceinline__ bool pixel(int j, int& m, int& b) const {
        m = mbase + j * 16 + frow;
        b = m / HWo;
        return m < M;
    }
    __device__ __forceinline__ int chan(int i) const { return cbase + i * 16 + fchunk * 4; }
};

// ... of the stencil-window kernels: an 8 x 32 (WIDE) or 16 x 16 patch of one image
template <bool WIDE>
struct PatchGeo {
    int bimg, y0, x0, cbase, Hin, Win, wq, frow, fchunk;       // wq: the wave's pixel quarter
    __device__ __forceinline__ bool pixel(int j, int& m, int& b) const {
        const int yy = y0 + (WIDE ? 2 * wq + (j >> 1) : 4 * wq + j), xx = x0 + (WIDE ? (j & 1) * 16 : 0) + frow;
        m = (bimg * Hin + yy) * Win + xx;
        b = bimg;
        return true;
    }
    __device__ __forceinline__ int chan(int i) const { return cbase + i * 16 + fchunk * 4; }
};

template <int BN, bool A_F32>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvParams p) {
    constexpr int WN = BN / 2;        // output channels per wave
    constexpr int MT = WN / 16;       // weight (A-operand) tiles per wave
    constexpr int PT = 4;             // pixel (B-operand) tiles per wave: 64 pixels
    constexpr int BROWS = BN / 32;    // weight rows staged per thread

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                          // pixels  [2][128][128 B]
    char* sB = smem + 2 * BM * 128;           // weights [2][BN][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    const int nwg = p.ntiles_m * p.ntiles_n;
    const int bid = xcd_remap(blockIdx.x, nwg);
    const int tn = bid % p.ntiles_n;
    const int tm = bid / p.ntiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int HWo = p.Hout * p.Wout;
    const int M = p.B * HWo;

    const int bz = blockIdx.y;   // batched-GEMM index (VAE attention); 0 otherwise
    const char* xbase = (const char*)p.x + (size_t)bz * p.batch_stride_x * (A_F32 ? 4 : 2);
    const uint16_t* wbase = p.w + (size_t)bz * p.batch_stride_w;

    // split-K range
    int kt_begin = 0, kt_end = p.ktiles_total;
    if (p.ksplit > 1) {
        int per = (p.ktiles_total + p.ksplit - 1) / p.ksplit;
        kt_begin = blockIdx.z * per;
        kt_end = min(p.ktiles_total, kt_begin + per);
        if (kt_begin >= kt_end) return;
    }

    // ---- per-thread staging geometry -------------------------------------------------------
    const int chunk = tid & 7;      // 16-byte chunk (8 channels) within the 64-channel K-step
    const int srow = tid >> 3;      // 0..31
    int a_iy0[4], a_ix0[4], a_boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + srow + 32 * i;
        if (m < M) {
            int b = m / HWo, r = m - b * HWo;
            int oy = r / p.Wout, ox = r - oy * p.Wout;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride - p.pad;
            a_boff[i] = b * p.Hin * p.Win;
        } else {
            a_iy0[i] = -(1 << 28);   // never valid
            a_ix0[i] = 0;
            a_boff[i] = 0;
        }
    }
    const int Heff = p.up ? 2 * p.Hin : p.Hin;
    const int Weff = p.up ? 2 * p.Win : p.Win;

    f32x4 acc[MT][PT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fchunk = lane >> 4;

    auto compute = [&](int buf) {
        const char* a = sA + buf * (BM * 128);
        const char* b = sB + buf * (BN * 128);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fw[MT], fx[PT];
            const int cc = 4 * s + fchunk;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                int r = wn * WN + i * 16 + frow;
                fw[i] = *(const bf16x8*)(b + r * 128 + ((cc ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                int r = wm * 64 + j * 16 + frow;
                fx[j] = *(const bf16x8*)(a + r * 128 + ((cc ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (A_F32) {
        // f32 activations (gradients, the raw residual stream): the pixel operand is register-staged with the bf16
        // conversion on the way into LDS (guide T14: loads of tile k+1 are issued before tile k is multiplied,
        // converted and written after); the bf16 weights go by LDS-DMA, which keeps more than half of the tile
        // bytes off the ds_write path.  Address generation is hoisted as in the LDS-DMA branch.
        constexpr unsigned OOB = 0x80000000u;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int lchunk = (lane & 7) ^ ((lane >> 3) & 7);
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, p.w_bytes, 0x00020000);
        int a_off[4];
        unsigned a_mask[4], b_base[BROWS], b_kill[BROWS];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a_off[i] = (a_boff[i] + a_iy0[i] * p.Win + a_ix0[i]) * (int)p.ldx + chunk * 8;
            unsigned mk = 0;
            for (int t = 0; t < p.KH * p.KW; ++t) {
                int ky = t / p.KW, kx = t - ky * p.KW;
                int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                if (iy >= 0 && iy < Heff && ix >= 0 && ix < Weff) mk |= 1u << t;
            }
            a_mask[i] = mk;
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            int n = n0 + srow + 32 * i;
            b_kill[i] = n < p.Cout ? 0u : OOB;
            b_base[i] = (unsigned)((n * p.Cin + lchunk * 8) * 2);
        }
        const float* xf = (const float*)xbase;
        float4 af[4][2];
        auto load_a = [&](int kt) {
            int tap = kt / p.ktiles_per_tap;
            int c0 = (kt - tap * p.ktiles_per_tap) * BK;
            int ky = tap / p.KW, kx = tap - ky * p.KW;
            bool cok = c0 + chunk * 8 < p.Cin;
            const int tapoff = (ky * p.Win + kx) * (int)p.ldx + c0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bool ok;
                int off;
                if (p.up == 0) {
                    ok = cok && ((a_mask[i] >> tap) & 1u);
                    off = a_off[i] + tapoff;
                } else {
                    int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                    ok = cok && iy >= 0 && iy < Heff && ix >= 0 && ix < Weff;
                    if (p.up == 2) ok = ok && !((iy | ix) & 1);
                    iy >>= 1; ix >>= 1;
                    off = (a_boff[i] + iy * p.Win + ix) * (int)p.ldx + c0 + chunk * 8;
                }
                float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
                if (ok) {
                    const float4* src = (const float4*)(xf + off);
                    f0 = src[0];
                    f1 = src[1];
                }
                af[i][0] = f0;
                af[i][1] = f1;
            }
        };
        auto store_a = [&](int buf) {
            char* a = sA + buf * (BM * 128);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = srow + 32 * i;
                uint4 v;
                v.x = pack_bf16x2(af[i][0].x, af[i][0].y);
                v.y = pack_bf16x2(af[i][0].z, af[i][0].w);
                v.z = pack_bf16x2(af[i][1].x, af[i][1].y);
                v.w = pack_bf16x2(af[i][1].z, af[i][1].w);
                *(uint4*)(a + r * 128 + ((chunk ^ (r & 7)) << 4)) = v;
            }
        };
        auto stage_b = [&](int buf, int kt) {
            int tap = kt / p.ktiles_per_tap;
            int c0 = (kt - tap * p.ktiles_per_tap) * BK;
            unsigned kill = (c0 + lchunk * 8 < p.Cin) ? 0u : OOB;
            asm volatile("" : "+v"(kill));
            const unsigned tapw = (unsigned)((tap * p.Cout * p.Cin + c0) * 2);
            char* bbase = sB + buf * (BN * 128) + wv * 1024;
#pragma unroll
            for (int i = 0; i < BROWS; ++i) {
                unsigned voff = (b_base[i] + tapw) | b_kill[i] | kill;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void_t*)(bbase + i * 4096), 16, voff, 0, 0, 0);
            }
        };
        load_a(kt_begin);
        stage_b(0, kt_begin);
        store_a(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int cur = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool more = (kt + 1 < kt_end);
            if (more) {
                load_a(kt + 1);
                stage_b(cur ^ 1, kt + 1);
            }
            compute(cur);
            if (more) store_a(cur ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // bf16 activations: LDS-DMA (global_load_lds_dwordx4) straight into the swizzled tile image -- no staging
        // registers and no ds_write traffic (register staging of a 128x160x64 tile needs 115 B/clk of ds_write
        // against ~80 B/clk available, so it caps the MFMA pipe at ~2/3).  The LDS destination of one wave
        // instruction is linear (8 rows x 128 B), so the XOR swizzle is applied to the per-lane SOURCE chunk
        // (guide rule 21); out-of-image / out-of-range lanes read a zero block instead.
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int rsw = (lane >> 3) & 7;                 // (row & 7) of every row this lane stages
        const int lchunk = (lane & 7) ^ rsw;             // logical 16-B chunk whose data lands in physical chunk lane&7
        // buffer descriptors (wave-uniform): an out-of-range offset makes the hardware deliver zeros, which is the
        // conv's zero padding, the ragged last pixel/channel tile and the Cin tail -- no branch, one DMA per row group
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, p.w_bytes, 0x00020000);
        constexpr unsigned OOB = 0x80000000u;
        // Address generation is hoisted out of the K loop: per staged row one byte offset of the tap-(0,0) source
        // pixel and a 9-bit mask of the taps that fall inside the image; per K step the tap contributes a
        // wave-uniform scalar offset.  (The naive per-tap decode cost ~35 VALU per DMA -- more issue cycles than
        // the MFMAs of the step.)
        unsigned a_base[4], a_mask[4], b_base[BROWS], b_kill[BROWS];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a_base[i] = (unsigned)(((a_boff[i] + a_iy0[i] * p.Win + a_ix0[i]) * (int)p.ldx + lchunk * 8) * 2);
            unsigned mk = 0;
            for (int t = 0; t < p.KH * p.KW; ++t) {
                int ky = t / p.KW, kx = t - ky * p.KW;
                int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                if (iy >= 0 && iy < Heff && ix >= 0 && ix < Weff) mk |= 1u << t;
            }
            a_mask[i] = mk;
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            int n = n0 + srow + 32 * i;
            b_kill[i] = n < p.Cout ? 0u : OOB;
            b_base[i] = (unsigned)((n * p.Cin + lchunk * 8) * 2);
        }
        auto stage = [&](int buf, int kt) {
            int tap = kt / p.ktiles_per_tap;
            int c0 = (kt - tap * p.ktiles_per_tap) * BK;
            int ky = tap / p.KW, kx = tap - ky * p.KW;
            bool cok = c0 + lchunk * 8 < p.Cin;
            char* abase = sA + buf * (BM * 128) + wv * 1024;
            char* bbase = sB + buf * (BN * 128) + wv * 1024;
            if (p.up == 0) {
                // branch-free: an invalid lane gets bit 31 OR-ed into its offset (>= num_records -> hardware zeros),
                // so every wave issues exactly the same number of DMA instructions per stage
                const unsigned tapoff = (unsigned)(((ky * p.Win + kx) * (int)p.ldx + c0) * 2);
                const unsigned tapw = (unsigned)((tap * p.Cout * p.Cin + c0) * 2);
                unsigned kill = cok ? 0u : OOB;
                asm volatile("" : "+v"(kill));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned inval = ((a_mask[i] >> tap) & 1u) - 1u;           // 0 if the tap is inside the image
                    unsigned voff = (a_base[i] + tapoff) | (inval & OOB) | kill;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(abase + i * 4096), 16, voff, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < BROWS; ++i) {
                    unsigned voff = (b_base[i] + tapw) | b_kill[i] | kill;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void_t*)(bbase + i * 4096), 16, voff, 0, 0, 0);
                }
                return;
            }
            // upsample / zero-insert gathers (3 + 3 layers): generic per-tap decode
            int c = c0 + lchunk * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                bool ok = cok && iy >= 0 && iy < Heff && ix >= 0 && ix < Weff;
                if (p.up == 2) ok = ok && !((iy | ix) & 1);
                iy >>= 1; ix >>= 1;
                unsigned voff = ok ? (unsigned)(((a_boff[i] + iy * p.Win + ix) * (int)p.ldx + c) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(abase + i * 4096), 16, voff, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < BROWS; ++i) {
                int n = n0 + srow + 32 * i;
                unsigned voff = (cok && n < p.Cout) ? (unsigned)(((tap * p.Cout + n) * p.Cin + c) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void_t*)(bbase + i * 4096), 16, voff, 0, 0, 0);
            }
        };
        stage(0, kt_begin);
        int cur = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's part of tile kt has landed
            __syncthreads();                                      // ... everyone's has, and compute(kt-1) is over
            if (kt + 1 < kt_end) stage(cur ^ 1, kt + 1);
            compute(cur);
            cur ^= 1;
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------
    // acc[i][j][r] = C[cout = n0 + wn*WN + i*16 + (lane>>4)*4 + r][pixel = m0 + wm*64 + j*16 + (lane&15)]
    float* y32 = p.y32 ? p.y32 + (size_t)bz * p.batch_stride_y32 : nullptr;
    uint16_t* y16 = p.y16 ? p.y16 + (size_t)bz * p.batch_stride_y16 : nullptr;
    const GemmGeo geo{m0 + wm * 64, n0 + wn * WN, M, HWo, frow, fchunk};
    if (p.ksplit > 1) conv_store_slab<MT, PT>(p, acc, geo, p.ws + (size_t)blockIdx.z * M * p.Cout);
    else if (p.epi == 1) conv_epilogue_geglu_fwd<MT, PT>(p, acc, geo);
    else if (p.epi == 2) conv_epilogue_geglu_bwd<MT, PT>(p, acc, geo);
    else if (p.gn_part == nullptr) {
        conv_epilogue<MT, PT, GemmGeo, 1>(p, acc, geo, y32, y16);      // (batches of one pixel fragment: this kernel lives on 2-3 workgroups per CU)
    } else {                                   // + GroupNorm statistics of the tile (epilogue_gn_stats; HWo % 128 == 0: host)
        float st[MT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i) { st[i][0] = 0.f; st[i][1] = 0.f; }
        conv_epilogue<MT, PT, GemmGeo, 1>(p, acc, geo, y32, y16, st);
        epilogue_gn_stats<MT>(p, st, frow, fchunk, n0 + wn * WN, (size_t)(m0 / 64) + wm);      // 64-pixel ranges, image-major
    }
}

// split-K second pass: fixed-order sum of the slabs + the fused epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvParams p, int M, int HWo) {
    const int q = p.Cout >> 2;
    const long total = (long)M * q;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int m = (int)(idx / q);
        const int c0 = (int)(idx - (long)m * q) * 4;
        float4 a = *(const float4*)(p.ws + (size_t)m * p.Cout + c0);
        for (int z = 1; z < p.ksplit; ++z) {
            float4 t = *(const float4*)(p.ws + ((size_t)z * M + m) * p.Cout + c0);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        float v[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
        if (p.bias) {
            float4 t = *(const float4*)(p.bias + c0);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (p.chan_add) {
            float4 t = *(const float4*)(p.chan_add + (size_t)(m / HWo) * p.ld_ca + c0);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (p.residual) {
            float4 t = *(const float4*)(p.residual + (size_t)m * p.ldr + c0);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (p.y32) *(float4*)(p.y32 + (size_t)m * p.ldy32 + c0) = make_float4(v[0], v[1], v[2], v[3]);
        if (p.y16) {
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)(p.y16 + (size_t)m * p.ldy16 + c0) = o;
        }
    }
}


// split-K second pass with the LayerNorm BACKWARD that consumes its result folded in (adap_conv2d_next_ln_bwd): the data-gradient
// contraction in front of a LayerNorm's backward (attention.py:267-269 norm1 / norm3 behind to_q|k|v / ff.net.0.proj) is a long-K
// GEMM that goes out split whenever the level has few pixel tiles -- its reduce pass already is a launch of its own that walks
// whole rows, so it can hand each row straight to the LayerNorm arithmetic instead of through memory and a further launch.  One
// wave per row, as ln_bwd_kernel (norms.hip); the slab sum, alpha / bias / residual and the LayerNorm arithmetic are those of
// splitk_reduce_kernel followed by ln_bwd_kernel in the same order: bit-identical to the two launches.  The contraction's own
// output (dy) is not stored.
struct LnBwdExt {
    const float* x; long ldx;          // the LayerNorm's input (saved by the forward)
    const float* gamma; const float* mean; const float* rstd;
    float* dx; long lddx; int accumulate;
    uint16_t* dx16; long lddx16;
};
#define SK_LN_MAXV 5    // float4 per lane: Cout <= 1280 (norms.hip LN_MAXV)

__global__ __launch_bounds__(256) void splitk_reduce_ln_bwd_kernel(ConvParams p, LnBwdExt e, int M) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int D = p.Cout, Q = D >> 2;
    const float mu = e.mean[row], rs = e.rstd[row];
    float dyh[SK_LN_MAXV][4], xh[SK_LN_MAXV][4];
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < SK_LN_MAXV; ++k) {
        const int q = lane + 64 * k;
        if (q < Q) {
            const int c0 = 4 * q;
            float4 a = *(const float4*)(p.ws + (size_t)row * D + c0);
            for (int z = 1; z < p.ksplit; ++z) {
                const float4 t = *(const float4*)(p.ws + ((size_t)z * M + row) * D + c0);
                a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
            }
            float ds[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
            if (p.bias) {
                const float4 t = *(const float4*)(p.bias + c0);
                ds[0] += t.x; ds[1] += t.y; ds[2] += t.z; ds[3] += t.w;
            }
            if (p.residual) {
                const float4 t = *(const float4*)(p.residual + (size_t)row * p.ldr + c0);
                ds[0] += t.x; ds[1] += t.y; ds[2] += t.z; ds[3] += t.w;
            }
            const float4 xv = *(const float4*)(e.x + row * e.ldx + c0);
            const float4 g = *(const float4*)(e.gamma + c0);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                xh[k][c] = (xs[c] - mu) * rs;
                dyh[k][c] = ds[c] * gs[c];
                sa += dyh[k][c];
                sb += dyh[k][c] * xh[k][c];
            }
        }
    }
    sa = wave_sum(sa) / D;
    sb = wave_sum(sb) / D;
#pragma unroll
    for (int k = 0; k < SK_LN_MAXV; ++k) {
        const int q = lane + 64 * k;
        if (q < Q) {
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = rs * (dyh[k][c] - sa - xh[k][c] * sb);
            float* dst = e.dx + row * e.lddx + 4 * q;
            if (e.accumulate) {
                const float4 t = *(const float4*)dst;
                o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
            }
            *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
            if (e.dx16) {
                uint2 w;
                w.x = pack_bf16x2(o[0], o[1]);
                w.y = pack_bf16x2(o[2], o[3]);
                *(uint2*)(e.dx16 + row * e.lddx16 + 4 * q) = w;
            }
        }
    }
}

// =============================================================================================
// Large-problem variant: 256 pixels x BN channels per workgroup, 8 waves (4 pixel quarters x 2 channel halves),
// bf16 activations only, THREE LDS stages filled by LDS-DMA with the prefetch two K steps ahead: the wait
// before a step is a COUNTED s_waitcnt vmcnt(loads of one stage) and the barrier is a raw s_barrier, so the
// DMA of the next tile stays in flight across it (guide section 5 "Pipelining across barriers", T3/T4).
// A 128 x 128 tile needs 64 B/clk/CU of operand traffic at the full MFMA rate -- the whole L2 bandwidth; the
// 256-wide tile halves that, and the two-step prefetch covers ~2 x 640 MFMA cycles of memory latency.
// =============================================================================================
#define BMB 256
#define WAIT_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt(int n) {      // n is wave-uniform; the immediate must be a literal
    switch (n) {
        WAIT_VMCNT_CASE(0) WAIT_VMCNT_CASE(1) WAIT_VMCNT_CASE(2) WAIT_VMCNT_CASE(3) WAIT_VMCNT_CASE(4) WAIT_VMCNT_CASE(5)
        WAIT_VMCNT_CASE(6) WAIT_VMCNT_CASE(7) WAIT_VMCNT_CASE(8) WAIT_VMCNT_CASE(9) WAIT_VMCNT_CASE(10) WAIT_VMCNT_CASE(11)
        WAIT_VMCNT_CASE(12) WAIT_VMCNT_CASE(13) WAIT_VMCNT_CASE(14) WAIT_VMCNT_CASE(15) WAIT_VMCNT_CASE(16) WAIT_VMCNT_CASE(17)
        WAIT_VMCNT_CASE(18) WAIT_VMCNT_CASE(19) WAIT_VMCNT_CASE(20) WAIT_VMCNT_CASE(21) WAIT_VMCNT_CASE(22) WAIT_VMCNT_CASE(23)
        WAIT_VMCNT_CASE(24) WAIT_VMCNT_CASE(25) WAIT_VMCNT_CASE(26) WAIT_VMCNT_CASE(27)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// BMT = 256 (8 waves, NST = 3: the large-problem variant above) or 128 (4 waves, NST = 4: problems with fewer
// workgroups than CUs, where occupancy is moot and the latency of one workgroup's K loop is everything -- three K
// steps of DMA in flight instead of one).
// ROLE: 0 = one barrier per K step; 1 / 2 = the early / late half of the ping-pong schedule (BMT = 256 only; see
// halo_body: waves w and w+4 share a SIMD and run half a step apart, so one of them is always in its MFMA phase).
// ONE_TAP: 1x1 / Linear (no tap decode, no validity masks in the staging path).
template <int BMT, int BN, int NST, int ROLE, bool ONE_TAP>
__device__ __forceinline__ void ring_body(const ConvParams& p) {
    constexpr int WN = BN / 2;
    constexpr int MT = WN / 16;
    constexpr int PT = 4;
    constexpr int RP = BMT / 4;                       // rows staged per pass (threads / 8)
    constexpr int A_STAGE = BMT * 128, B_STAGE = BN * 128;
    constexpr int BPASS = (BN + RP - 1) / RP;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + NST * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int wm = (BMT == 256) ? (wv & 3) : (wv & 1);
    const int wn = (BMT == 256) ? (wv >> 2) : (wv >> 1);

    const int nwg = p.ntiles_m * p.ntiles_n;
    const int bid = xcd_remap(blockIdx.x, nwg);
    const int tn = bid % p.ntiles_n;
    const int tm = bid / p.ntiles_n;
    const int m0 = tm * BMT, n0 = tn * BN;
    const int HWo = p.Hout * p.Wout;
    const int M = p.B * HWo;

    int kt_begin = 0, kt_end = p.ktiles_total;
    if (p.ksplit > 1) {
        int per = (p.ktiles_total + p.ksplit - 1) / p.ksplit;
        kt_begin = blockIdx.z * per;
        kt_end = min(p.ktiles_total, kt_begin + per);
    }

    const int srow = tid >> 3;                        // 0..RP-1
    const int rsw = (lane >> 3) & 7;
    const int lchunk = (lane & 7) ^ rsw;
    int a_iy0[4], a_ix0[4], a_boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + srow + RP * i;
        if (m < M) {
            int b = m / HWo, r = m - b * HWo;
            int oy = r / p.Wout, ox = r - oy * p.Wout;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride - p.pad;
            a_boff[i] = b * p.Hin * p.Win;
        } else {
            a_iy0[i] = -(1 << 28);
            a_ix0[i] = 0;
            a_boff[i] = 0;
        }
    }
    const int Heff = p.up ? 2 * p.Hin : p.Hin;
    const int Weff = p.up ? 2 * p.Win : p.Win;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    // weight-row passes this wave takes part in (a pass = 64 rows, a wave = 8 of them): 4 A + nB B loads per stage
    int nB = 0;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) nB += (RP * i + 8 * wv < BN) ? 1 : 0;

    // hoisted address generation (see conv_gemm_kernel): per row a base offset + a tap-validity mask
    unsigned a_base[4], a_mask[4], b_base[BPASS], b_kill[BPASS];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_base[i] = (unsigned)(((a_boff[i] + a_iy0[i] * p.Win + a_ix0[i]) * (int)p.ldx + lchunk * 8) * 2);
        unsigned mk = 0;
        for (int t = 0; t < p.KH * p.KW; ++t) {
            int ky = t / p.KW, kx = t - ky * p.KW;
            int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
            if (iy >= 0 && iy < Heff && ix >= 0 && ix < Weff) mk |= 1u << t;
        }
        a_mask[i] = mk;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        int n = n0 + srow + RP * i;
        b_kill[i] = n < p.Cout ? 0u : OOB;
        b_base[i] = (unsigned)((n * p.Cin + lchunk * 8) * 2);
    }
    auto stage = [&](int buf, int kt) {
        char* abase = sA + buf * A_STAGE + wv * 1024;
        char* bbase = sB + buf * B_STAGE + wv * 1024;
        if (ONE_TAP) {
            const int c0 = kt * BK;
            unsigned kill = (c0 + lchunk * 8 < p.Cin) ? 0u : OOB;
            asm volatile("" : "+v"(kill));
            const unsigned off = (unsigned)(c0 * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned voff = (a_base[i] + off) | ((a_mask[i] & 1u) ? 0u : OOB) | kill;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(abase + i * (RP * 128)), 16, voff, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                if (RP * i + 8 * wv < BN) {               // wave-uniform
                    unsigned voff = (b_base[i] + off) | b_kill[i] | kill;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void_t*)(bbase + i * (RP * 128)), 16, voff, 0, 0, 0);
                }
            }
            return;
        }
        int tap = kt / p.ktiles_per_tap;
        int c0 = (kt - tap * p.ktiles_per_tap) * BK;
        int ky = tap / p.KW, kx = tap - ky * p.KW;
        bool cok = c0 + lchunk * 8 < p.Cin;
        {
            // branch-free (see conv_gemm_kernel): the counted vmcnt below relies on every wave issuing exactly
            // 4 + nB DMA instructions per stage
            const unsigned tapoff = (unsigned)(((ky * p.Win + kx) * (int)p.ldx + c0) * 2);
            const unsigned tapw = (unsigned)((tap * p.Cout * p.Cin + c0) * 2);
            unsigned kill = cok ? 0u : OOB;
            asm volatile("" : "+v"(kill));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned inval = ((a_mask[i] >> tap) & 1u) - 1u;
                unsigned voff = (a_base[i] + tapoff) | (inval & OOB) | kill;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(abase + i * (RP * 128)), 16, voff, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                if (RP * i + 8 * wv < BN) {               // wave-uniform
                    unsigned voff = (b_base[i] + tapw) | b_kill[i] | kill;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void_t*)(bbase + i * (RP * 128)), 16, voff, 0, 0, 0);
                }
            }
        }
    };

    f32x4 acc[MT][PT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int fchunk = lane >> 4;

    // fragment byte offsets, computed once (the K half s = 1 is the same chunk position XOR 4 -> offset XOR 64)
    int woff[MT], xoff[PT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int r = wn * WN + i * 16 + frow;
        woff[i] = r * 128 + ((fchunk ^ (r & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        int r = wm * 64 + j * 16 + frow;
        xoff[j] = r * 128 + ((fchunk ^ (r & 7)) << 4);
    }
    bf16x8 fw[2][MT], fx[2][PT];
    auto load_frags = [&](int buf) {
        const char* a = sA + buf * A_STAGE;
        const char* b = sB + buf * B_STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int i = 0; i < MT; ++i) fw[s2][i] = *(const bf16x8*)(b + (woff[i] ^ (s2 * 64)));
#pragma unroll
            for (int j = 0; j < PT; ++j) fx[s2][j] = *(const bf16x8*)(a + (xoff[j] ^ (s2 * 64)));
        }
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[s2][i], fx[s2][j], acc[i][j], 0, 0, 0);
    };

    const int ntl = kt_end - kt_begin;
    const int L = 4 + nB;                               // DMA instructions this wave issues per stage
    // diagnostic timeline (tools/gemm_timeline.py; p.clk NULL in production): 100 MHz real-time stamps of this workgroup's
    // start, first operand tile landed, K loop done, epilogue stores issued
    unsigned long long* stamps = p.clk ? p.clk + 4 * ((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.z) : nullptr;
    if (stamps && tid == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
        if (i < ntl) stage(i, kt_begin + i);
    int buf = 0;
    bool pending = false;
    for (int t = 0; t < ntl; ++t) {
        // tile t has landed once only the stages issued after it (tiles t+1 .. t+NST-2) can still be in flight
        int ahead = ntl - 1 - t;
        if (ahead > NST - 2) ahead = NST - 2;
        wait_vmcnt(ahead * L);
        __builtin_amdgcn_s_barrier();          // X: tile t visible to every wave; every wave has read tile t-1
        asm volatile("" ::: "memory");
        if (stamps && t == 0 && tid == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
        auto issue = [&]() {
            if (t + NST - 1 < ntl) {
                int nb = buf + NST - 1;
                if (nb >= NST) nb -= NST;
                stage(nb, kt_begin + t + NST - 1); // refills the buffer tile t-1 was multiplied from
            }
        };
        if (ROLE == 0) {
            issue();
            load_frags(buf);
            mfmas();
        } else if (ROLE == 1) {
            issue();
            load_frags(buf);
            __builtin_amdgcn_s_barrier();      // Y
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
            mfmas();
            __builtin_amdgcn_s_setprio(0);
        } else {
            if (pending) {
                __builtin_amdgcn_s_setprio(1);
                mfmas();
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_s_barrier();      // Y
            asm volatile("" ::: "memory");
            issue();
            load_frags(buf);
            pending = true;
        }
        if (++buf == NST) buf = 0;
    }
    if (ROLE == 2 && pending) mfmas();         // the late half's last step (no barrier: the early half is done)
    if (stamps && tid == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();

    {
        const GemmGeo geo{m0 + wm * 64, n0 + wn * WN, M, HWo, frow, fchunk};
        if (p.ksplit > 1) conv_store_slab<MT, PT>(p, acc, geo, p.ws + (size_t)blockIdx.z * M * p.Cout);
        else if (p.epi == 1) conv_epilogue_geglu_fwd<MT, PT>(p, acc, geo);
        else if (p.epi == 2) conv_epilogue_geglu_bwd<MT, PT>(p, acc, geo);
        else if (p.gn_part == nullptr) {
            conv_epilogue<MT, PT, GemmGeo, 2>(p, acc, geo, p.y32, p.y16);
        } else {                               // + GroupNorm statistics of the tile (epilogue_gn_stats; HWo % BMT == 0: host)
            float st[MT][2];
#pragma unroll
            for (int i = 0; i < MT; ++i) { st[i][0] = 0.f; st[i][1] = 0.f; }
            conv_epilogue<MT, PT, GemmGeo, 2>(p, acc, geo, p.y32, p.y16, st);
            epilogue_gn_stats<MT>(p, st, frow, fchunk, n0 + wn * WN, (size_t)(m0 / 64) + wm);  // 64-pixel ranges, image-major
        }
    }
    if (stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamps[3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int BMT, int BN, int NST, bool ONE_TAP>
__global__ __launch_bounds__(BMT * 2) void conv_gemm_ring_kernel(ConvParams p) {
    // ping-pong halves measured on the 256-row variant: no gain (these GEMMs are bound by the L2 -> LDS operand feed,
    // ~7.8 TB/s aggregate at 52 KB per K step per CU, not by the per-step schedule), so the plain schedule is used
    constexpr bool PP = false;
    if (!PP) ring_body<BMT, BN, NST, 0, ONE_TAP>(p);
    else if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) ring_body<BMT, BN, NST, 1, ONE_TAP>(p);   // waves 0-3
    else ring_body<BMT, BN, NST, 2, ONE_TAP>(p);                                                              // waves 4-7
}

static int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// the LayerNorm backward folded into the CURRENT adap_conv2d_nhwc call's split-K reduce (adap_conv2d_next_ln_bwd), or inactive
static thread_local LnBwdExt g_ln_bwd_cur = {};
static thread_local bool g_ln_bwd_cur_on = false;
static thread_local int g_ln_bwd_last = 0;

static thread_local LnBwdExt g_ln_bwd_next = {};
static thread_local bool g_ln_bwd_next_on = false;
extern "C" int adap_conv2d_next_ln_bwd(const float* x, long ldx, const float* gamma, const float* mean, const float* rstd, float* dx,
                                       long lddx, int accumulate, void* dx16, long lddx16) {
    ADAP_REQUIRE(x && gamma && mean && rstd && dx, ADAP_ERR_SHAPE, "conv2d_next_ln_bwd: null pointer");
    ADAP_REQUIRE(ldx % 4 == 0 && lddx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)gamma % 16) == 0 &&
                 (!dx16 || (lddx16 % 4 == 0 && ((uintptr_t)dx16 % 8) == 0)), ADAP_ERR_ALIGN, "conv2d_next_ln_bwd: alignment");
    g_ln_bwd_next = LnBwdExt{x, ldx, gamma, mean, rstd, dx, lddx, accumulate, (uint16_t*)dx16, lddx16};
    g_ln_bwd_next_on = true;
    return ADAP_OK;
}
extern "C" int adap_conv2d_last_ln_bwd(void) { return g_ln_bwd_last; }

static void launch_reduce(const ConvParams& p, hipStream_t stream) {
    const int M = p.B * p.Hout * p.Wout;
    if (g_ln_bwd_cur_on && !p.chan_add && p.epi == 0 && p.Cout % 4 == 0 && p.Cout <= 256 * SK_LN_MAXV) {
        hipLaunchKernelGGL(splitk_reduce_ln_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, p, g_ln_bwd_cur, M);
        g_ln_bwd_last = 1;
        return;
    }
    long total = (long)M * (p.Cout >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p, M, p.Hout * p.Wout);
}

template <int BMT, int BN, int NST, bool ONE_TAP>
static int launch_ring_t(const ConvParams& p, hipStream_t stream) {
    size_t lds = (size_t)NST * (BMT + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_gemm_ring_kernel<BMT, BN, NST, ONE_TAP>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(p.ntiles_m * p.ntiles_n, 1, p.ksplit);
    hipLaunchKernelGGL((conv_gemm_ring_kernel<BMT, BN, NST, ONE_TAP>), grid, dim3(BMT * 2), lds, stream, p);
    if (p.ksplit > 1) launch_reduce(p, stream);
    return adap_check_launch("conv_gemm_ring");
}

template <int BMT, int BN, int NST>
static int launch_ring(const ConvParams& p, hipStream_t stream) {
    if (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.up == 0) return launch_ring_t<BMT, BN, NST, true>(p, stream);
    return launch_ring_t<BMT, BN, NST, false>(p, stream);
}

// =============================================================================================
// 3x3 / stride 1 / pad 1 convolution with an LDS-staged STENCIL WINDOW (the ResBlock and VAE ResnetBlock convs,
// i.e. ~95 % of the conv FLOPs).  A workgroup owns an 8 x 32 pixel patch of one image and BN output channels.  Per
// 64-channel slice of Cin it DMAs the (8+2) x (32+2) halo window ONCE into LDS and runs all nine taps out of it --
// each tap is a 1x1 contraction whose pixel operand is the same window read at a shifted position -- while the nine
// [BN x 64] weight slices stream through a 3-deep ring.  Compared with gathering one tap per K step this cuts the
// activation traffic L2 -> LDS from 9x to 1.33x; the gather variant was bound by exactly that feed rate (measured:
// 2600-4300 cycles per K step against 1024 of MFMA, HBM traffic already compulsory-only).
// =============================================================================================
// patch shapes: WIDE = 8 x 32 pixels (window 10 x 34 = 340 slots), !WIDE = 16 x 16 (window 18 x 18 = 324 slots) for the
// UNet's 16 x 16 level, whose rows are too short for the wide patch.  Either way 256 pixels and <= 384 staged slots.
#define HALO_PASSES 6                            // 6 x 64 slots >= 340
template <bool WIDE>
struct HaloShape {
    static constexpr int TH = WIDE ? 8 : 16, TW = WIDE ? 32 : 16;
    static constexpr int W = TW + 2, SLOTS = (TH + 2) * W;
};
template <int BN, int ROLE, bool WIDE>
__device__ __forceinline__ void halo_body(const ConvParams& p) {
    constexpr int HALO_TH = HaloShape<WIDE>::TH, HALO_TW = HaloShape<WIDE>::TW, HALO_W = HaloShape<WIDE>::W;
    constexpr int HALO_SLOTS = HaloShape<WIDE>::SLOTS;
    constexpr int WN = BN / 2;
    constexpr int MT = WN / 16;
    constexpr int PT = 4;
    constexpr int A_STAGE = HALO_PASSES * 64 * 128;      // 48 KB
    constexpr int B_STAGE = BN * 128;
    constexpr int BPASS = (BN + 63) / 64;
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                       // [2][384 slots][128 B]
    char* sB = smem + 2 * A_STAGE;         // [3][BN][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv & 3, wn = wv >> 2;

    const int tiles_x = p.Win / HALO_TW, tiles_y = p.Hin / HALO_TH;
    const int nwg = p.ntiles_m * p.ntiles_n;

    const int nchunks_total = p.ktiles_per_tap;        // 64-channel slices of Cin
    int ch_begin = 0, ch_end = nchunks_total;
    if (p.ksplit > 1) {
        int per = (nchunks_total + p.ksplit - 1) / p.ksplit;
        ch_begin = blockIdx.z * per;
        ch_end = min(nchunks_total, ch_begin + per);
    }
    const int nch = ch_end - ch_begin;
    if (nch <= 0) return;

    const int srow = tid >> 3;                          // 0..63
    // LDS image of the window: slot (hr, hc) -> 128-B row hr*34 + hc, its eight 16-B chunks XOR-swizzled by the window
    // COLUMN (hc & 7).  Keying on the column (not the slot index) makes a tap's row shift a plain +34*128*ky bytes
    // (an instruction immediate), so only the three kx variants of a fragment address live in registers.  16 lanes of
    // a fragment read 16 consecutive columns of one row: every key appears twice, on opposite bank halves.
    const int wchunk = (lane & 7) ^ ((lane >> 3) & 7);  // weights: rows of 8*wv + lane/8, key = row & 7

    // (A persistent variant -- one workgroup per CU walking the tiles as one unbroken stream of K steps, the next
    // tile's window and weights prefetched across the tile boundary -- was built and measured: 5-15 % SLOWER than one
    // tile per workgroup, because the epilogue's stores share the vmcnt counter with the DMA loads and force a full
    // drain anyway, and the hardware dispatcher balances tiles better.  Removed.)
    struct Tile {
        unsigned a_base[HALO_PASSES];   // halo slots staged by this thread: s = 64*i + srow; bit 31 set = zero fill
        unsigned b_base[BPASS];         // (offsets stay below 2^31, so adding the chunk offset keeps the bit)
        int bimg, y0, x0, n0;
    };
    auto decode = [&](int tile_id, Tile& c) {
        const int bid = xcd_remap(tile_id, nwg);
        const int tn = bid % p.ntiles_n;
        int tm = bid / p.ntiles_n;
        const int tx = tm % tiles_x;
        tm /= tiles_x;
        const int ty = tm % tiles_y;
        c.bimg = tm / tiles_y;
        c.y0 = ty * HALO_TH;
        c.x0 = tx * HALO_TW;
        c.n0 = tn * BN;
#pragma unroll
        for (int i = 0; i < HALO_PASSES; ++i) {
            int sidx = 64 * i + srow;
            int hr = sidx / HALO_W, hc = sidx - hr * HALO_W;
            int iy = c.y0 - 1 + hr, ix = c.x0 - 1 + hc;
            bool ok = sidx < HALO_SLOTS && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
            int lchunk = (lane & 7) ^ (hc & 7);
            c.a_base[i] = ok ? (unsigned)((((c.bimg * p.Hin + iy) * p.Win + ix) * (int)p.ldx + lchunk * 8) * 2) : OOB;
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            int n = c.n0 + srow + 64 * i;
            c.b_base[i] = n < p.Cout ? (unsigned)((n * p.Cin + wchunk * 8) * 2) : OOB;
        }
    };
    int nB = 0;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) nB += (64 * i + 8 * wv < BN) ? 1 : 0;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    auto stage_a = [&](const Tile& c, int buf, int chunk) {     // HALO_PASSES DMA instructions per wave
        const int c0 = chunk * BK;                               // Cin % 64 == 0 (choose_halo): no ragged K slice
        char* base = sA + buf * A_STAGE + wv * 1024;
#pragma unroll
        for (int i = 0; i < HALO_PASSES; ++i) {
            unsigned voff = c.a_base[i] + (unsigned)(c0 * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(base + i * 8192), 16, voff, 0, 0, 0);
        }
    };
    auto stage_b = [&](const Tile& c, int buf, int chunk, int tap) {    // nB DMA instructions per wave
        const int c0 = chunk * BK;
        const unsigned tapw = (unsigned)((tap * p.Cout * p.Cin + c0) * 2);
        char* base = sB + buf * B_STAGE + wv * 1024;
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            if (64 * i + 8 * wv < BN) {
                unsigned voff = c.b_base[i] + tapw;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void_t*)(base + i * 8192), 16, voff, 0, 0, 0);
            }
        }
    };

    f32x4 acc[MT][PT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int fchunk = lane >> 4;
    // LDS byte offsets of every fragment this lane will ever read, computed ONCE: the nine taps are unrolled below, so
    // the ring slot (tap % 3), the tap's window shift and the K half (s) fold into immediates / these registers and
    // the per-step instruction stream is DMA issue + 16 ds_read_b128 + 32 MFMA, no address arithmetic.
    //   window: slot of this lane's pixel in fragment j at tap (ky,kx) = (2*wm + (j>>1) + ky) * 34 + (j&1)*16 + frow + kx
    //   the s = 1 half is the same 16-B chunk position XOR 4  ->  byte offset XOR 64
    int aoff[3][PT];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            int hc = (WIDE ? (j & 1) * 16 : 0) + frow + kx;
            int sl = (WIDE ? 2 * wm + (j >> 1) : 4 * wm + j) * HALO_W + hc;
            aoff[kx][j] = sl * 128 + ((fchunk ^ (hc & 7)) << 4);
        }
    int woff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int r = wn * WN + i * 16 + frow;
        woff[i] = r * 128 + ((fchunk ^ (r & 7)) << 4);
    }

    const int M = p.B * p.Hin * p.Win;
    auto epilogue = [&](int c_bimg, int c_y0, int c_x0, int c_n0) {
        const PatchGeo<WIDE> geo{c_bimg, c_y0, c_x0, c_n0 + wn * WN, p.Hin, p.Win, wm, frow, fchunk};
        if (p.ksplit > 1) {
            conv_store_slab<MT, PT>(p, acc, geo, p.ws + (size_t)blockIdx.z * M * p.Cout);
        } else if (p.gn_part == nullptr) {
            conv_epilogue<MT, PT>(p, acc, geo, p.y32, p.y16);
        } else {
            float st[MT][2];
#pragma unroll
            for (int i = 0; i < MT; ++i) { st[i][0] = 0.f; st[i][1] = 0.f; }
            conv_epilogue<MT, PT>(p, acc, geo, p.y32, p.y16, st);
            epilogue_gn_stats<MT>(p, st, frow, fchunk, c_n0 + wn * WN,
                                  ((size_t)c_bimg * (tiles_x * tiles_y) + (c_y0 / HALO_TH) * tiles_x + c_x0 / HALO_TW) * 4 + wm);
        }
    };

    // ---- the step stream.  A step = (chunk, tap); the nine taps of a chunk are unrolled (ring slot = tap % 3 because
    // 9 % 3 == 0).  B(step+2) is issued at each step, the next chunk's window at tap 0 after it; "step+2" and "next
    // chunk" run on into the next tile of this workgroup.
    //
    // PP (ping-pong): waves 0-3 (channel half 0) and waves 4-7 (channel half 1) share the four SIMDs pairwise.  With a
    // single barrier per step both waves of a SIMD are always in the same phase -- DMA issue, LDS reads, then MFMA --
    // and the matrix pipe idles through the first two.  Here a step has two barriers (X, Y) and the halves run half a
    // step apart: between X and Y half 0 issues its DMAs and reads its fragments while half 1 runs the MFMAs of the
    // PREVIOUS step out of registers; between Y and the next X they swap.  Every SIMD always has one wave in its MFMA
    // phase.  Ring slots and window buffers are only overwritten by DMAs issued after the X that follows their last
    // reader, exactly as without PP, because half 1 finishes reading step t before X(t+1).
    if ((int)blockIdx.x >= nwg) return;
    bf16x8 fw[2][MT], fx[2][PT];
    auto load_frags = [&](const char* a, int tap) {
        const char* b = sB + (tap % 3) * B_STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int i = 0; i < MT; ++i) fw[s2][i] = *(const bf16x8*)(b + (woff[i] ^ (s2 * 64)));
#pragma unroll
            for (int j = 0; j < PT; ++j)
                fx[s2][j] = *(const bf16x8*)(a + (tap / 3) * (HALO_W * 128) + (aoff[tap % 3][j] ^ (s2 * 64)));
        }
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[s2][i], fx[s2][j], acc[i][j], 0, 0, 0);
    };

    // one role per instantiation of this body (ROLE 0: single-barrier schedule; 1: early half; 2: late half), so that
    // each gets its own straight-line schedule and register allocation; all roles execute the same barriers
    Tile cur;
    decode(blockIdx.x, cur);
    unsigned long long clk0 = 0, rt0 = 0;
    if (p.clk && tid == 0) {                // in-kernel clock probe (MI355X_MICROARCH.md "DVFS give-back", item 6)
        clk0 = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    stage_a(cur, 0, ch_begin);
    stage_b(cur, 0, ch_begin, 0);
    stage_b(cur, 1, ch_begin, 1);
    bool pending = false;  // late half: fragments of the previous step are in registers, MFMAs not yet issued
    for (int ci = 0; ci < nch; ++ci) {
        const bool last_chunk = ci + 1 == nch;
        const char* a = sA + (ci & 1) * A_STAGE;          // window of chunk ci lives in sA[ci & 1]
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // DMA instructions that may still be in flight while this step's operands are complete: B(step+1), and the
            // next chunk's window during taps 1-2
            const bool more_steps = tap < 8 || !last_chunk;
            if ((tap == 1 || tap == 2) && !last_chunk) {
                if (nB == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            } else if (more_steps) {
                if (nB == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                       // X: operands of this step are in LDS
            asm volatile("" ::: "memory");
            auto issue = [&]() {
                if (ROLE == 0 && (p.dbg & 1)) return;
                const int nb = (tap + 2) % 3;
                if (tap + 2 < 9) stage_b(cur, nb, ch_begin + ci, tap + 2);
                else if (!last_chunk) stage_b(cur, nb, ch_begin + ci + 1, tap + 2 - 9);
                if (tap == 0 && !last_chunk) stage_a(cur, (ci + 1) & 1, ch_begin + ci + 1);
            };
            if (ROLE == 0) {
                issue();
                if (!(p.dbg & 2)) {
                    load_frags(a, tap);
                    mfmas();
                }
            } else if (ROLE == 1) {
                issue();
                load_frags(a, tap);
                __builtin_amdgcn_s_barrier();                   // Y
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_setprio(1);
                mfmas();
                __builtin_amdgcn_s_setprio(0);
            } else {
                if (pending) {
                    __builtin_amdgcn_s_setprio(1);
                    mfmas();
                    __builtin_amdgcn_s_setprio(0);
                }
                __builtin_amdgcn_s_barrier();                   // Y
                asm volatile("" ::: "memory");
                issue();
                load_frags(a, tap);
                pending = true;
            }
        }
    }
    if (ROLE == 2) mfmas();             // the late half's last step: no barrier any more (the early half is done)
    if (p.clk && tid == 0) {
        p.clk[2 * (blockIdx.x + (size_t)gridDim.x * blockIdx.z)] = __builtin_amdgcn_s_memtime() - clk0;
        p.clk[2 * (blockIdx.x + (size_t)gridDim.x * blockIdx.z) + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
    if (ROLE != 0 || !(p.dbg & 4)) epilogue(cur.bimg, cur.y0, cur.x0, cur.n0);
}

template <int BN, bool PP, bool WIDE>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(ConvParams p) {
    if (!PP) halo_body<BN, 0, WIDE>(p);
    else if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) halo_body<BN, 1, WIDE>(p);   // waves 0-3
    else halo_body<BN, 2, WIDE>(p);                                                              // waves 4-7
}

// ---------------------------------------------------------------------------------------------
// Two-workgroups-per-CU variant of the stencil-window kernel.  The 512-thread kernel above owns a whole CU (144 KB of
// LDS), so a tile's prologue (first window + weights in flight) and epilogue (192 KB of stores per tile, all CUs at
// once) overlap with nothing.  Here a workgroup is FOUR waves (one per SIMD) on the same 256 px x 128 ch tile, each wave
// 64 px x 128 ch (128 accumulator registers), K slices of 32 channels (64-B LDS rows): 2 x 24 KB window + 3 x 8 KB
// weight ring = 72 KB, so two workgroups share a CU and one's prologue / epilogue / barrier waits run under the
// other's MFMAs.  L2 -> LDS bytes per flop are unchanged (same tile), LDS reads drop from 16 to 12 ds_read_b128 per
// 32 MFMAs (the weight fragments are shared by 4 pixel tiles... the window fragments by 8 channel tiles).
// 64-B rows: a ds_read_b128 fragment covers 16 consecutive rows x 4 chunks = 1 KB; the hardware's 16-lane groups mix
// rows {0-3, 12-15} of one chunk with rows {4-11} of the next, which collide 2-way on (row mod 4); XOR-ing the chunk
// with 2 where bit 2 of the row key (window column / weight row) is set separates them for every alignment.
template <int BN, bool WIDE>
__device__ __forceinline__ void win32_body(const ConvParams& p) {
    constexpr int HALO_TH = HaloShape<WIDE>::TH, HALO_TW = HaloShape<WIDE>::TW, HALO_W = HaloShape<WIDE>::W;
    constexpr int HALO_SLOTS = HaloShape<WIDE>::SLOTS;
    constexpr int MT = BN / 16;                           // every wave holds all BN channels of its 64 pixels
    constexpr int PT = 4;
    constexpr int A_STAGE = HALO_PASSES * 64 * 64;        // 384 slots x 64 B = 24 KB
    constexpr int B_STAGE = BN * 64;
    constexpr int BPASS = (BN + 63) / 64;
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                       // [2][384 slots][64 B]
    char* sB = smem + 2 * A_STAGE;         // [3][BN][64 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..3 = pixel quarter

    const int tiles_x = p.Win / HALO_TW, tiles_y = p.Hin / HALO_TH;
    const int nwg = p.ntiles_m * p.ntiles_n;
    if ((int)blockIdx.x >= nwg) return;

    int ch_begin = 0, ch_end = p.ktiles_per_tap;          // split-K plan is in 64-channel units (host)
    if (p.ksplit > 1) {
        int per = (p.ktiles_per_tap + p.ksplit - 1) / p.ksplit;
        ch_begin = blockIdx.z * per;
        ch_end = min(p.ktiles_per_tap, ch_begin + per);
    }
    if (ch_end <= ch_begin) return;
    const int c32_begin = 2 * ch_begin, nch = 2 * (ch_end - ch_begin);     // 32-channel slices

    // staging: a DMA instruction moves 1 KB = 16 rows x 64 B; thread -> row (tid >> 2) of each 64-row pass, chunk tid & 3
    const int srow = tid >> 2;
    const int bid = xcd_remap(blockIdx.x, nwg);
    const int tn = bid % p.ntiles_n;
    int tm = bid / p.ntiles_n;
    const int tx = tm % tiles_x;
    tm /= tiles_x;
    const int ty = tm % tiles_y;
    const int bimg = tm / tiles_y;
    const int y0 = ty * HALO_TH, x0 = tx * HALO_TW, n0 = tn * BN;
    unsigned a_base[HALO_PASSES], b_base[BPASS];
#pragma unroll
    for (int i = 0; i < HALO_PASSES; ++i) {
        int sidx = 64 * i + srow;
        int hr = sidx / HALO_W, hc = sidx - hr * HALO_W;
        int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
        bool ok = sidx < HALO_SLOTS && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        int lchunk = (lane & 3) ^ ((hc >> 1) & 2);
        a_base[i] = ok ? (unsigned)((((bimg * p.Hin + iy) * p.Win + ix) * (int)p.ldx + lchunk * 8) * 2) : OOB;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        int r = srow + 64 * i;
        int n = n0 + r;
        int wchunk = (lane & 3) ^ ((r >> 1) & 2);
        b_base[i] = (r < BN && n < p.Cout) ? (unsigned)((n * p.Cin + wchunk * 8) * 2) : OOB;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    auto stage_a = [&](int buf, int c32) {                  // HALO_PASSES DMA instructions per wave
        char* base = sA + buf * A_STAGE + wv * 1024;
#pragma unroll
        for (int i = 0; i < HALO_PASSES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(base + i * 4096), 16,
                                                     a_base[i] + (unsigned)(c32 * 64), 0, 0, 0);
    };
    auto stage_b = [&](int buf, int c32, int tap) {         // BPASS DMA instructions per wave
        const unsigned tapw = (unsigned)((tap * p.Cout * p.Cin + c32 * 32) * 2);
        char* base = sB + buf * B_STAGE + wv * 1024;
#pragma unroll
        for (int i = 0; i < BPASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void_t*)(base + i * 4096), 16, b_base[i] + tapw, 0, 0, 0);
    };
    static_assert(BN % 64 == 0, "every wave stages BPASS weight pieces");
    constexpr int nB = BPASS;

    f32x4 acc[MT][PT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int fchunk = lane >> 4;
    // fragment offsets: the key depends on the window column only, (hc + 16) has the same key, so of the 4 pixel
    // fragments and 9 taps only the three kx variants live in registers; everything else is an immediate
    int aoff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        int hc = frow + kx;
        int sl = (WIDE ? 2 * wv : 4 * wv) * HALO_W + hc;
        aoff[kx] = sl * 64 + ((fchunk ^ ((hc >> 1) & 2)) << 4);
    }
    const int woff = frow * 64 + ((fchunk ^ ((frow >> 1) & 2)) << 4);      // + i * 1024 (16 rows keep the key)

    bf16x8 fw[MT], fx[PT];
    for (int ci = 0; ci < nch; ++ci) {
        if (ci == 0) {
            stage_a(0, c32_begin);
            stage_b(0, c32_begin, 0);
            stage_b(1, c32_begin, 1);
        }
        const bool last_chunk = ci + 1 == nch;
        const char* a = sA + (ci & 1) * A_STAGE;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bool more_steps = tap < 8 || !last_chunk;
            if ((tap == 1 || tap == 2) && !last_chunk) {
                if (nB == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (nB == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            } else if (more_steps) {
                if (nB == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (nB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!(p.dbg & 1)) {
                const int nb = (tap + 2) % 3;
                if (tap + 2 < 9) stage_b(nb, c32_begin + ci, tap + 2);
                else if (!last_chunk) stage_b(nb, c32_begin + ci + 1, tap + 2 - 9);
                if (tap == 0 && !last_chunk) stage_a((ci + 1) & 1, c32_begin + ci + 1);
            }
            if (p.dbg & 2) continue;
            const char* b = sB + (tap % 3) * B_STAGE;
            const char* at = a + (tap / 3) * (HALO_W * 64) + aoff[tap % 3];
#pragma unroll
            for (int j = 0; j < PT; ++j)
                fx[j] = *(const bf16x8*)(at + (WIDE ? (j >> 1) * (HALO_W * 64) + (j & 1) * (16 * 64) : j * (HALO_W * 64)));
#pragma unroll
            for (int i = 0; i < MT; ++i) fw[i] = *(const bf16x8*)(b + woff + i * 1024);
            // All twelve fragment reads are issued before the first MFMA: left alone, the compiler reloads ONE weight
            // register quad just in time and exposes the LDS latency eight times per step.  The empty asm pins the
            // first eight fragments (and, by its memory clobber, the issue of the other four) ahead of the first half
            // of the MFMAs; the second half waits only for the rest.
            static_assert(MT == 8, "fragment pinning below is written for the 128-channel tile");
            asm volatile("" : "+v"(fx[0]), "+v"(fx[1]), "+v"(fx[2]), "+v"(fx[3]), "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]),
                         "+v"(fw[3]) :: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(fw[4]), "+v"(fw[5]), "+v"(fw[6]), "+v"(fw[7]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 4; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
        }
    }

    if (p.dbg & 4) return;
    const int M = p.B * p.Hin * p.Win;
    const PatchGeo<WIDE> geo{bimg, y0, x0, n0, p.Hin, p.Win, wv, frow, fchunk};
    if (p.ksplit > 1) conv_store_slab<MT, PT>(p, acc, geo, p.ws + (size_t)blockIdx.z * M * p.Cout);
    else conv_epilogue<MT, PT, PatchGeo<WIDE>, 1>(p, acc, geo, p.y32, p.y16);
}

template <int BN, bool WIDE>
__global__ __launch_bounds__(256, 2) void conv3x3_win32_kernel(ConvParams p) {
    win32_body<BN, WIDE>(p);
}

template <int BN, bool WIDE>
static int launch_win32(const ConvParams& p, hipStream_t stream) {
    size_t lds = 2 * HALO_PASSES * 64 * 64 + 3 * BN * 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv3x3_win32_kernel<BN, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr_set = true;
    }
    const int nwg = p.ntiles_m * p.ntiles_n;
    hipLaunchKernelGGL((conv3x3_win32_kernel<BN, WIDE>), dim3(nwg, 1, p.ksplit), dim3(256), lds, stream, p);
    if (p.ksplit > 1) launch_reduce(p, stream);
    return adap_check_launch("conv3x3_win32");
}

template <int BN, bool WIDE>
static int launch_halo(const ConvParams& p, hipStream_t stream) {
    constexpr bool PP = BN == 128;      // ping-pong halves (see halo_body); the 160-wide tile spills with them
    size_t lds = 2 * HALO_PASSES * 64 * 128 + 3 * BN * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv3x3_halo_kernel<BN, PP, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr_set = true;
    }
    // one workgroup per CU at a time (144 KB of LDS each), one tile per workgroup
    const int nwg = p.ntiles_m * p.ntiles_n;
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN, PP, WIDE>), dim3(nwg, 1, p.ksplit), dim3(512), lds, stream, p);
    if (p.ksplit > 1) launch_reduce(p, stream);
    return adap_check_launch("conv3x3_halo");
}

template <int BN, bool A_F32>
static int launch(const ConvParams& p, int nbatch, hipStream_t stream) {
    size_t lds = 2 * BM * 128 + 2 * BN * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_gemm_kernel<BN, A_F32>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(p.ntiles_m * p.ntiles_n, nbatch, p.ksplit);
    hipLaunchKernelGGL((conv_gemm_kernel<BN, A_F32>), grid, dim3(256), lds, stream, p);
    if (p.ksplit > 1) launch_reduce(p, stream);
    return adap_check_launch("conv_gemm");
}

static int choose_bn(int Cout) {
    // channel tile: 160 divides 320/640/960/1280/...; 128 for the VAE's powers of two; 64 for tiny Cout
    if (Cout <= 64) return 64;
    if (Cout % 160 == 0) return 160;
    return 128;
}

// the 256-pixel, 3-stage variant: bf16 activations, one problem, enough pixels and K steps to fill its pipeline
static bool choose_big(long M, int Cout, int ktiles_total, int x_dtype, int nbatch, int up) {
    return up == 0 && x_dtype == 1 && nbatch == 1 && Cout > 64 && M >= 4096 && ktiles_total >= 5;
}

// patch shape of the stencil-window kernel for an H x W image: 1 = 8 x 32 (wide), 2 = 16 x 16 (square), 0 = neither tiles it
static int halo_shape(int H, int W) {
    if (H % 8 == 0 && W % 32 == 0) return 1;
    if (H % 16 == 0 && W % 16 == 0) return 2;
    return 0;
}

// the stencil-window variant: 3x3 / stride 1 / pad 1 on bf16 activations whose image tiles into 256-pixel patches
static bool choose_halo(int Hin, int Win, int Hout, int Wout, int Cin, int Cout, int KH, int KW, int stride, int pad, int up,
                        int x_dtype, int nbatch) {
    return KH == 3 && KW == 3 && stride == 1 && pad == 1 && up == 0 && x_dtype == 1 && nbatch == 1 && Hout == Hin &&
           Wout == Win && halo_shape(Hin, Win) != 0 && Cout > 64 && Cin >= 64 && Cin % 64 == 0;
}

// channel tile of the stencil-window kernel: the ping-pong schedule exists for the 128-wide tile only (the 160-wide one
// has no registers left for a second fragment set), and it is worth more than a partly empty last tile costs:
// 128 whenever it wastes < 10 % of the channel tiles (640, 960, 1280, 1920, 2560 ...), else 160 (320)
static int choose_bn_halo(int Cout) {
    int bn = choose_bn(Cout);
    if (bn == 160) {
        int t128 = (Cout + 127) / 128;
        if (t128 * 128 * 10 <= Cout * 11) bn = 128;
    }
    return bn;
}

// Channel tile and split factor of the stencil-window kernel, from a small cost model calibrated on dependent chains of the
// UNet's ResBlock convs (tools/conv3_probe.py, in us): a launch costs ~12 (boundary, prologue, epilogue) plus its K steps
// (9 taps x Cin slices per workgroup) at 0.49 per step for the 128-wide ping-pong tile and 0.78 for the 160-wide one, times
// the ROUNDS the grid needs on the chip's CUs -- a split that pushes the grid just past one round (280 workgroups on 256
// CUs) nearly doubles the launch, which is what the old "ceil(256 / tiles)" rule did at 32^2 and 16^2 -- plus, when split,
// the reduce pass: its own boundary and (splits + 1.5) x the f32 output through L2 at ~5 TB/s.
static double halo_split_penalty() {          // tuning: extra us charged to a split plan (ADAP_HALO_SPLIT_PENALTY)
    static const double v = getenv("ADAP_HALO_SPLIT_PENALTY") ? atof(getenv("ADAP_HALO_SPLIT_PENALTY")) : 0.0;
    return v;
}

static void choose_halo_plan(int B, int H, int W, int Cin, int Cout, int* bn_out, int* ks_out) {
    const int nchunks = (Cin + BK - 1) / BK;
    const long M = (long)B * H * W;
    const int ncu = num_cus();
    double best = 1e30;
    int best_bn = choose_bn_halo(Cout), best_ks = 1;
    for (int bn = 128; bn <= 160; bn += 32) {
        const long blocks = (long)B * (H * W / 256) * ((Cout + bn - 1) / bn);
        const double step_us = bn == 128 ? 0.49 : 0.78;
        for (int ks = 1; ks <= 16 && ks <= nchunks; ++ks) {
            const int per = (nchunks + ks - 1) / ks;
            if ((nchunks + per - 1) / per != ks) continue;                 // (the dispatcher would round it to this anyway)
            const long rounds = (blocks * ks + ncu - 1) / ncu;
            double t = 4.0 + (8.0 + per * 9 * step_us) * (double)rounds;
            if (ks > 1) t += 4.0 + (double)M * Cout * 4.0 * (ks + 1.5) / 5.0e6 + halo_split_penalty();
            if (t < best) { best = t; best_bn = bn; best_ks = ks; }
        }
    }
    *bn_out = best_bn;
    *ks_out = best_ks;
}

static int choose_ksplit_halo(int B, int H, int W, int Cin, int Cout) {
    int bn, ks;
    choose_halo_plan(B, H, W, Cin, Cout, &bn, &ks);
    return ks;
}

// split-K plan: a launch needs >> 256 workgroups to fill the chip; the UNet's 32x32 .. 8x8 levels have few pixel
// tiles but very long K (up to 9*2560), so split K until there are ~2 workgroups per CU (1 per CU for the big tile).
// `g_ksplit_scale` (percent, adap_conv_ksplit_scale): the targets are for a launch that has the chip to itself; a caller that keeps
// two streams busy (the micro-batch lanes) asks for fewer, longer workgroups -- the other stream fills the CUs, and the slab
// traffic and the reduce launch go away (measured under lanes: 35 % -> -0.2 ms per micro-batch; alone: +0.8 ms).  The workspace
// query always sizes for 100 %, the largest plan, so sizes cached by a caller stay valid whatever the scale is later.
static int g_ksplit_scale = -1;
static int ksplit_scale() {
    if (g_ksplit_scale < 0) {
        const char* e = getenv("ADAP_KSPLIT_SCALE");
        int v = e ? atoi(e) : 100;
        g_ksplit_scale = v < 0 ? 0 : (v > 100 ? 100 : v);
    }
    return g_ksplit_scale;
}

extern "C" int adap_conv_ksplit_scale(int percent) {
    const int prev = ksplit_scale();
    if (percent >= 0) g_ksplit_scale = percent > 100 ? 100 : percent;
    return prev;
}

static int choose_ksplit(long M, int Cout, int ktiles_total, bool big, int scale = -1) {
    int bn = choose_bn(Cout);
    int bm = big ? BMB : BM;
    long blocks = ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn);
    long enough = big ? 200 : 384, target = big ? 256 : 512;
    if (scale < 0) scale = ksplit_scale();
    if (scale <= 0) return 1;
    enough = enough * scale / 100;
    target = target * scale / 100;
    // Few workgroups: besides idle CUs, every workgroup then streams a large slice of (cold, HBM-resident) weights
    // through ONE CU's ~10 B/clk fetch path.  Spread the K loop over more workgroups -- but only for long-K layers:
    // for short K the slab reduce pass costs what the split saves (measured both ways on the UNet's 1x1 layers).
    if (blocks >= enough || ktiles_total < 24) return 1;
    int ks = (int)((target + blocks - 1) / blocks);
    int cap = ktiles_total / 4;
    if (ks > cap) ks = cap;
    if (ks > 16) ks = 16;
    return ks < 1 ? 1 : ks;
}

// floats of split-K workspace adap_conv2d_nhwc needs for this problem with ksplit = 0 (auto); 0 = none.
// (sized for either activation dtype: the larger of the two plans)
extern "C" long adap_conv2d_workspace_floats(int B, int Hout, int Wout, int Cin, int Cout, int KH, int KW) {
    long M = (long)B * Hout * Wout;
    int kt = KH * KW * ((Cin + BK - 1) / BK);
    int ks0 = choose_ksplit(M, Cout, kt, false, 100);
    int ks1 = choose_big(M, Cout, kt, 1, 1, 0) ? choose_ksplit(M, Cout, kt, true, 100) : 1;
    int ks = ks0 > ks1 ? ks0 : ks1;
    if (KH == 3 && KW == 3 && halo_shape(Hout, Wout) != 0 && Cout > 64 && Cin >= 64) {
        int ks2 = choose_ksplit_halo(B, Hout, Wout, Cin, Cout);
        if (ks2 > ks) ks = ks2;
    }
    return ks > 1 ? (long)ks * M * Cout : 0;
}

static bool narrow_tiles_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAP_NARROW_TILES");
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}

// which kernel the last adap_conv2d_nhwc call of this thread dispatched to (bench.py's per-kernel roofline):
// 1000*variant + BN, variant 0 = conv_gemm_kernel f32 activations, 1 = conv_gemm_kernel bf16, 2 = conv_gemm_ring_kernel<256,.,3>,
// 3 = conv_gemm_ring_kernel<128,.,4>, 4 = conv3x3_halo_kernel, 5 = conv3x3_win32_kernel
// diagnostic: a device buffer of 2 * (workgroups of the next stencil-window launches) uint64, or NULL (production)
static unsigned long long* g_clock_probe = nullptr;
extern "C" int adap_conv2d_set_clock_probe(void* buf) {
    g_clock_probe = (unsigned long long*)buf;
    return ADAP_OK;
}

static thread_local int g_last_variant = -1;
extern "C" int adap_conv2d_last_variant(void) { return g_last_variant; }

// the fused-GEGLU entry points below hand their extra operands to adap_conv2d_nhwc through this (same thread, next call)
struct EpiExt { int epi; uint16_t* z16; long ldz16; const uint16_t* h16; long ldh16; };
static thread_local EpiExt g_epi_ext = {0, nullptr, 0, nullptr, 0};

// GroupNorm statistics from the NEXT adap_conv2d_nhwc call's epilogue (one shot, same thread)
static thread_local float* g_gn_next = nullptr;
static thread_local int g_gn_next_cpg = 0;
static thread_local int g_gn_last_chunks = 0;
extern "C" int adap_conv2d_next_gn_partial(float* partial, int channels_per_group) {
    ADAP_REQUIRE(!partial || (channels_per_group > 0 && ((uintptr_t)partial % 8) == 0), ADAP_ERR_SHAPE,
                 "conv2d_next_gn_partial: cpg %d", channels_per_group);
    g_gn_next = partial;
    g_gn_next_cpg = channels_per_group;
    return ADAP_OK;
}
extern "C" int adap_conv2d_last_gn_chunks(void) { return g_gn_last_chunks; }

static thread_local int g_force_kind = 0, g_force_bn = 0;
extern "C" int adap_conv2d_debug_force(int kind, int bn) {
    ADAP_REQUIRE(kind >= 0 && kind <= 3 && (bn == 0 || bn == 64 || bn == 128 || bn == 160), ADAP_ERR_UNSUPPORTED,
                 "conv2d_debug_force: kind %d bn %d", kind, bn);
    g_force_kind = kind;
    g_force_bn = bn;
    return ADAP_OK;
}

extern "C" int adap_conv2d_nhwc(
    const void* x, int x_dtype, long ldx,
    const void* w_packed,
    const float* bias, const float* chan_add, long ld_ca,
    const float* residual, long ldr,
    float* y32, long ldy32, void* y16, long ldy16,
    int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
    int KH, int KW, int stride, int pad, int up,
    float alpha, int ksplit, float* splitk_ws,
    int nbatch, long bs_x, long bs_w, long bs_y32, long bs_y16,
    void* stream) {
    // the one-shot GroupNorm-statistics request is consumed by THIS call whatever happens to it: were it left armed behind a
    // failed argument check, a later conv of this thread would write its records into a buffer the caller has freed by then
    float* const gn_next = g_gn_next;
    const int gn_next_cpg = g_gn_next_cpg;
    g_gn_next = nullptr;
    g_gn_last_chunks = 0;
    // ... and so is the one-shot LayerNorm-backward request (adap_conv2d_next_ln_bwd): live for this call only, on every exit
    struct LnBwdScope {
        LnBwdScope() {
            g_ln_bwd_cur = g_ln_bwd_next;
            g_ln_bwd_cur_on = g_ln_bwd_next_on;
            g_ln_bwd_next_on = false;
            g_ln_bwd_last = 0;
        }
        ~LnBwdScope() { g_ln_bwd_cur_on = false; }
    } ln_scope;
    ADAP_REQUIRE(x && w_packed && (y32 || y16), ADAP_ERR_SHAPE, "conv2d: null pointer");
    ADAP_REQUIRE(x_dtype == 0 || x_dtype == 1, ADAP_ERR_UNSUPPORTED, "conv2d: x_dtype %d", x_dtype);
    ADAP_REQUIRE(B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Cin > 0 && Cout > 0,
                 ADAP_ERR_SHAPE, "conv2d: non-positive dims");
    ADAP_REQUIRE(Cin % 8 == 0, ADAP_ERR_ALIGN, "conv2d: Cin=%d must be a multiple of 8", Cin);
    ADAP_REQUIRE(Cout % 4 == 0, ADAP_ERR_ALIGN, "conv2d: Cout=%d must be a multiple of 4", Cout);
    ADAP_REQUIRE(ldx % 8 == 0 && ldx >= Cin, ADAP_ERR_ALIGN, "conv2d: ldx=%ld (Cin=%d)", ldx, Cin);
    ADAP_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0, ADAP_ERR_ALIGN,
                 "conv2d: x / w must be 16-byte aligned");
    ADAP_REQUIRE(!y32 || (ldy32 % 4 == 0 && ldy32 >= Cout && ((uintptr_t)y32 % 16) == 0),
                 ADAP_ERR_ALIGN, "conv2d: y32 alignment");
    ADAP_REQUIRE(!y16 || (ldy16 % 4 == 0 && ldy16 >= Cout && ((uintptr_t)y16 % 8) == 0),
                 ADAP_ERR_ALIGN, "conv2d: y16 alignment");
    ADAP_REQUIRE(!chan_add || (ld_ca % 4 == 0 && ((uintptr_t)chan_add % 16) == 0), ADAP_ERR_ALIGN,
                 "conv2d: chan_add alignment");
    ADAP_REQUIRE(!bias || ((uintptr_t)bias % 16) == 0, ADAP_ERR_ALIGN, "conv2d: bias alignment");
    ADAP_REQUIRE(!residual || (ldr % 4 == 0 && ((uintptr_t)residual % 16) == 0), ADAP_ERR_ALIGN,
                 "conv2d: residual alignment");
    ADAP_REQUIRE((KH == 1 && KW == 1) || (KH == 3 && KW == 3), ADAP_ERR_UNSUPPORTED,
                 "conv2d: kernel %dx%d", KH, KW);
    ADAP_REQUIRE(stride == 1 || stride == 2, ADAP_ERR_UNSUPPORTED, "conv2d: stride %d", stride);
    ADAP_REQUIRE(up >= 0 && up <= 2, ADAP_ERR_UNSUPPORTED, "conv2d: up %d", up);
    ADAP_REQUIRE(ksplit >= 0 && ksplit <= 64, ADAP_ERR_UNSUPPORTED, "conv2d: ksplit %d", ksplit);
    ADAP_REQUIRE(nbatch >= 1, ADAP_ERR_SHAPE, "conv2d: nbatch");
    long M = (long)B * Hout * Wout;
    ADAP_REQUIRE(M < (1L << 31) && (long)B * Hin * Win < (1L << 31), ADAP_ERR_SHAPE, "conv2d: too many pixels");

    ConvParams p;
    p.x = x; p.ldx = ldx; p.w = (const uint16_t*)w_packed; p.bias = bias; p.chan_add = chan_add; p.ld_ca = ld_ca;
    p.residual = residual; p.ldr = ldr; p.y32 = y32; p.ldy32 = ldy32; p.y16 = (uint16_t*)y16; p.ldy16 = ldy16;
    p.B = B; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Hout = Hout; p.Wout = Wout; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.up = up;
    p.ktiles_per_tap = (Cin + BK - 1) / BK;
    p.ktiles_total = KH * KW * p.ktiles_per_tap;
    p.epi = g_epi_ext.epi; p.z16 = g_epi_ext.z16; p.ldz16 = g_epi_ext.ldz16; p.h16 = g_epi_ext.h16; p.ldh16 = g_epi_ext.ldh16;
    p.gn_part = nullptr;
    p.gn_cpg = 0;
    const bool halo = choose_halo(Hin, Win, Hout, Wout, Cin, Cout, KH, KW, stride, pad, up, x_dtype, nbatch);
    const bool big = !halo && choose_big(M, Cout, p.ktiles_total, x_dtype, nbatch, up);
    const int ksplit_units = halo ? p.ktiles_per_tap : p.ktiles_total;      // the halo kernel splits over Cin slices
    if (ksplit == 0) {
        if (!(nbatch == 1 && splitk_ws)) ksplit = 1;
        else ksplit = halo ? choose_ksplit_halo(B, Hin, Win, Cin, Cout) : choose_ksplit(M, Cout, p.ktiles_total, big);
    }
    p.ksplit = ksplit < ksplit_units ? ksplit : ksplit_units;
    {   // every split must own at least one unit (an empty split would leave its slab unwritten)
        int per = (ksplit_units + p.ksplit - 1) / p.ksplit;
        p.ksplit = (ksplit_units + per - 1) / per;
    }
    ADAP_REQUIRE(p.ksplit == 1 || (splitk_ws && nbatch == 1 && ((uintptr_t)splitk_ws % 16) == 0), ADAP_ERR_SHAPE,
                 "conv2d: split-K needs a workspace (adap_conv2d_workspace_floats) and nbatch == 1");
    p.ws = splitk_ws;
    {
        long xb = ((long)B * Hin * Win - 1) * ldx + Cin, wb = (long)KH * KW * Cout * Cin;
        xb *= (x_dtype == 0 ? 4 : 2);
        wb *= 2;
        ADAP_REQUIRE(xb < (1L << 31) && wb < (1L << 31), ADAP_ERR_SHAPE, "conv2d: operand larger than 2 GiB");
        p.x_bytes = (unsigned)xb;
        p.w_bytes = (unsigned)wb;
    }
    p.ntiles_m = (int)((M + (big ? BMB : BM) - 1) / (big ? BMB : BM));
    p.alpha = alpha;
    p.batch_stride_x = bs_x; p.batch_stride_w = bs_w; p.batch_stride_y32 = bs_y32; p.batch_stride_y16 = bs_y16;
    {
        static int dbg = -1;
        if (dbg < 0) {
            const char* e = getenv("ADAP_CONV_DEBUG");
            dbg = e ? atoi(e) : 0;
        }
        p.dbg = dbg;
        p.clk = g_clock_probe;
    }
    hipStream_t s = (hipStream_t)stream;

    int bn = choose_bn(Cout);
    if (p.epi == 1) bn = Cout % 128 == 0 ? 128 : 64;       // a wave must hold value / gate tile PAIRS: 4 or 2 tiles per wave
    if (halo) {
        int ks_plan;
        choose_halo_plan(B, Hin, Win, Cin, Cout, &bn, &ks_plan);
        if (g_force_bn && g_force_bn != 64) bn = g_force_bn;               // diagnostic override (tools/conv3_probe.py)
    }
    bool use_big = big;
    if (!halo && nbatch == 1 && up == 0 && p.ksplit == 1 && p.ktiles_total >= 4 && narrow_tiles_enabled()) {
        // too few workgroups for 256 CUs and K too short for split-K to pay: trade tile size for workgroups.
        // 256x160 -> 128x160 -> 128x64 until there are >= 200 of them (these layers are latency-, not MFMA-bound)
        long t160 = (Cout + bn - 1) / bn, t64 = (Cout + 63) / 64;
        if (use_big && ((M + BMB - 1) / BMB) * t160 < 200) {
            use_big = false;
            p.ntiles_m = (int)((M + BM - 1) / BM);
        }
        if (!use_big && ((M + BM - 1) / BM) * t160 < 200 && ((M + BM - 1) / BM) * t64 > ((M + BM - 1) / BM) * t160) bn = 64;
    }
    if (!halo && g_force_kind != 0 && nbatch == 1 && up == 0 && x_dtype == 1) {      // diagnostic override (tools/gemm_probe.py)
        if (g_force_bn) bn = g_force_bn;
        if (g_force_kind == 2 && bn == 64) bn = 128;          // no 256 x 64 ring variant
        use_big = g_force_kind == 2;
        p.ntiles_m = (int)((M + (use_big ? BMB : BM) - 1) / (use_big ? BMB : BM));
        p.ntiles_n = (Cout + bn - 1) / bn;
        if (g_force_kind == 2) {
            g_last_variant = 2000 + bn;
            if (bn == 160) return launch_ring<256, 160, 3>(p, s);
            return launch_ring<256, 128, 3>(p, s);
        } else if (g_force_kind == 3) {
            g_last_variant = 3000 + bn;
            if (bn == 160) return launch_ring<128, 160, 4>(p, s);
            if (bn == 128) return launch_ring<128, 128, 4>(p, s);
            return launch_ring<128, 64, 4>(p, s);
        } else {
            g_last_variant = 1000 + bn;
            if (bn == 160) return launch<160, false>(p, nbatch, s);
            if (bn == 128) return launch<128, false>(p, nbatch, s);
            return launch<64, false>(p, nbatch, s);
        }
    }
    if (use_big && bn == 64) {                              // (no 256 x 64 ring variant)
        use_big = false;
        p.ntiles_m = (int)((M + BM - 1) / BM);
    }
    p.ntiles_n = (Cout + bn - 1) / bn;
    if (halo) {
        g_last_variant = 4000 + bn;
        p.ntiles_m = B * (Hin * Win / 256);
        // The four-wave kernel (two workgroups per CU) overlaps prologue + epilogue, a large share of a short-K tile: in
        // isolation +10 % on 128->128 @512^2 and +7 % on 640->640 @32^2 (<= 4 slices of 64 channels per workgroup), -6 %
        // on 512->512 @128^2 (long K: the ping-pong kernel's steady state is better).  Inside the training step, where
        // the VAE encode of the next micro-batch already shares the CUs from a second stream, the same rule measured 1 %
        // SLOWER end to end (110.4 vs 111.5 img/s, two interleaved runs each), so it stays opt-in:
        // ADAP_WIN32 = 0 never (default), 1 for <= 4 slices per workgroup, 2 always.
        static int win32 = -1;
        if (win32 < 0) {
            const char* e = getenv("ADAP_WIN32");
            win32 = e ? atoi(e) : 0;
        }
        const int slices_per_wg = (p.ktiles_per_tap + p.ksplit - 1) / p.ksplit;
        if (bn == 128 && (win32 == 2 || (win32 == 1 && slices_per_wg <= 4))) {
            g_last_variant = 5000 + bn;
            if (halo_shape(Hin, Win) == 1) return launch_win32<128, true>(p, s);
            return launch_win32<128, false>(p, s);
        }
        // GroupNorm statistics in the epilogue: unsplit 128-wide tiles whose channel quads do not straddle groups
        if (gn_next && p.ksplit == 1 && bn == 128 && p.epi == 0 && Cout % 128 == 0 && gn_next_cpg * 32 == Cout &&
            gn_next_cpg % 4 == 0 && 128 % gn_next_cpg == 0) {
            p.gn_part = gn_next;
            p.gn_cpg = gn_next_cpg;
            g_gn_last_chunks = Hin * Win / 64;
        }
        if (halo_shape(Hin, Win) == 1) {
            if (bn == 160) return launch_halo<160, true>(p, s);
            return launch_halo<128, true>(p, s);
        }
        if (bn == 160) return launch_halo<160, false>(p, s);
        return launch_halo<128, false>(p, s);
    }
    // GroupNorm statistics in the epilogue (adap_conv2d_next_gn_partial): 128-channel tiles of an unsplit, unbatched call whose
    // pixel tiles do not straddle images and whose channel quads do not straddle groups
    const bool gn_ok = gn_next && p.ksplit == 1 && nbatch == 1 && bn == 128 && p.epi == 0 && Cout % 128 == 0 &&
                       gn_next_cpg * 32 == Cout && gn_next_cpg % 4 == 0 && 128 % gn_next_cpg == 0;
    auto arm_gn = [&](int tile_px) {
        if (gn_ok && (Hout * Wout) % tile_px == 0) {
            p.gn_part = gn_next;
            p.gn_cpg = gn_next_cpg;
            g_gn_last_chunks = Hout * Wout / 64;
        }
    };
    if (use_big) {
        g_last_variant = 2000 + bn;
        static int big_nst = -1;                 // ADAP_RING_BIG_NST (tuning): 2 stages = 104 KB of LDS instead of all 160
        if (big_nst < 0) {
            const char* e = getenv("ADAP_RING_BIG_NST");
            big_nst = e ? atoi(e) : 3;
        }
        if (big_nst == 2) {
            if (bn == 160) return launch_ring<256, 160, 2>(p, s);
            arm_gn(256);
            return launch_ring<256, 128, 2>(p, s);
        }
        if (bn == 160) return launch_ring<256, 160, 3>(p, s);
        arm_gn(256);
        return launch_ring<256, 128, 3>(p, s);
    }
    // few workgroups (< 1.2 per CU): the 4-deep LDS-DMA ring hides the K-loop latency that occupancy cannot
    if (x_dtype == 1 && nbatch == 1 && up == 0 && p.ktiles_total >= 4 &&
        (long)p.ntiles_m * p.ntiles_n * p.ksplit <= 300) {
        g_last_variant = 3000 + bn;
        // ADAP_RING_NST (tuning): a shallower ring leaves room for a second workgroup per CU (LDS: (128 + BN) x 128 B per
        // stage) -- with two micro-batches in flight (MicroBatchLanes) another stream's workgroups can then share the CU
        static int nst = -1;
        if (nst < 0) {
            const char* e = getenv("ADAP_RING_NST");
            nst = e ? atoi(e) : 4;
        }
        if (nst == 2) {
            if (bn == 160) return launch_ring<128, 160, 2>(p, s);
            if (bn == 128) { arm_gn(128); return launch_ring<128, 128, 2>(p, s); }
            return launch_ring<128, 64, 2>(p, s);
        }
        if (nst == 3) {
            if (bn == 160) return launch_ring<128, 160, 3>(p, s);
            if (bn == 128) { arm_gn(128); return launch_ring<128, 128, 3>(p, s); }
            return launch_ring<128, 64, 3>(p, s);
        }
        if (bn == 160) return launch_ring<128, 160, 4>(p, s);
        if (bn == 128) { arm_gn(128); return launch_ring<128, 128, 4>(p, s); }
        return launch_ring<128, 64, 4>(p, s);
    }
    g_last_variant = (x_dtype == 0 ? 0 : 1000) + bn;
    if (bn == 128) arm_gn(BM);
    if (x_dtype == 0) {
        if (bn == 160) return launch<160, true>(p, nbatch, s);
        if (bn == 128) return launch<128, true>(p, nbatch, s);
        return launch<64, true>(p, nbatch, s);
    } else {
        if (bn == 160) return launch<160, false>(p, nbatch, s);
        if (bn == 128) return launch<128, false>(p, nbatch, s);
        return launch<64, false>(p, nbatch, s);
    }
}

// ---------------------------------------------------------------------------------------------
// conv3x3 on an RGB image (the VAE encoder's conv_in, model.py:426, 468: 3 -> ch at full resolution).  Through the general path
// the three channels are padded to 8 and every tap is a K step of 64 with 8 live columns: nine steps, 8x the matrix work, 357 us
// for an output that takes ~100 us to write.  Here the whole 3 x 3 x 3 patch of a pixel is ONE K step: k = 3 * tap + channel,
// 27 of 32 columns live.  No LDS: a lane builds its fragment of the pixel operand from the f32 image itself (8 scalar loads per
// 16-pixel fragment, L1 / L2 hits: the image is 12.6 MB) and its weight fragments from the general [9][Cout][8] pack; one
// MFMA 16x16x32 per accumulator; the shared epilogue (bias, f32 / bf16 out, GroupNorm statistics records) does the rest, which
// is the kernel's time: it is bound by the 537 MB of f32 output.  A wave owns 64 consecutive pixels of a row x all Cout <= 128
// channels; 256 threads = 256 pixels per workgroup.
// ---------------------------------------------------------------------------------------------
#define RGB_TILES 4                                  // 256-pixel tiles per workgroup (the weight image is staged once)
template <int MT>
__global__ __launch_bounds__(256) void conv3x3_rgb_kernel(ConvParams p) {
    constexpr int PT = 4;
    constexpr int ROWB = MT * 64 + 16;                 // bytes per staged pixel row (+16: a quad column's 16 rows spread over banks)
    __shared__ __attribute__((aligned(16))) uint16_t sW[MT * 16 * 32];     // [Cout][32] bf16, k = 3 * tap + channel (27 live)
    __shared__ __attribute__((aligned(16))) char sStage[4 * 16 * ROWB];    // per wave: 16 pixel rows of the output tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fchunk = lane >> 4;
    const int HWo = p.Hout * p.Wout, M = p.B * HWo;
    const float* __restrict__ x = (const float*)p.x;
    // the weight image out of the general [9][Cout][8] pack: 32 columns per output channel
    for (int idx = tid; idx < MT * 16 * 32; idx += 256) {
        const int r = idx >> 5, k = idx & 31, t = k / 3, c = k - 3 * t;
        sW[idx] = (k < 27 && r < p.Cout) ? p.w[((size_t)t * p.Cout + r) * p.Cin + c] : (uint16_t)0;
    }
    // this lane's eight k columns: k = 8 * fchunk + e -> (tap, channel); k >= 27 is padding
    int koff[8];                                       // element offset of the tap's pixel relative to the output pixel, + channel
    int kdy[8], kdx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * fchunk + e, t = k < 27 ? k / 3 : 4;
        kdy[e] = t / 3 - 1;
        kdx[e] = t - 3 * (t / 3) - 1;
        koff[e] = (kdy[e] * p.Win + kdx[e]) * (int)p.ldx + (k < 27 ? k - 3 * t : 0);
    }
    const bool klive_hi = fchunk < 3;                  // columns 24 .. 31: only 24, 25, 26 are live
    __syncthreads();
    bf16x8 fw[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) fw[i] = *(const bf16x8*)(sW + (i * 16 + frow) * 32 + fchunk * 8);
    for (int tile = 0; tile < RGB_TILES; ++tile) {
        const int m0 = (blockIdx.x * RGB_TILES + tile) * 256 + wv * 64;
        if (m0 >= M) break;                            // wave-uniform
        bf16x8 fx[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int m = m0 + j * 16 + frow;
            const int mm = m < M ? m : M - 1;
            const int b = mm / HWo, r = mm - b * HWo, oy = r / p.Wout, ox = r - oy * p.Wout;
            const float* px = x + ((size_t)(b * p.Hin + oy) * p.Win + ox) * p.ldx;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int iy = oy + kdy[e], ix = ox + kdx[e];
                const bool ok = (klive_hi || e < 3) && m < M && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
                const float v = ok ? px[koff[e]] : 0.f;
                fx[j][e] = (__bf16)v;
            }
        }
        f32x4 acc[MT][PT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (p.y16 != nullptr || m0 + 64 > M) {         // bf16 output or a ragged last tile: the shared epilogue
            const GemmGeo geo{m0, 0, M, HWo, frow, fchunk};
            if (p.gn_part == nullptr) {
                conv_epilogue<MT, PT, GemmGeo, 2>(p, acc, geo, p.y32, p.y16);
            } else {
                float st[MT][2];
#pragma unroll
                for (int i = 0; i < MT; ++i) { st[i][0] = 0.f; st[i][1] = 0.f; }
                conv_epilogue<MT, PT, GemmGeo, 2>(p, acc, geo, p.y32, p.y16, st);
                epilogue_gn_stats<MT>(p, st, frow, fchunk, 0, (size_t)(m0 / 64));
            }
            continue;
        }
        // ---- f32 output, whole tile: this kernel is nothing but its stores (537 MB at 512 x 512).  An accumulator quad is 16 bytes
        // of one pixel, so a store instruction of the shared epilogue writes 64-byte pieces of 16 different pixel rows (measured:
        // 231 us; with the statistics' vmcnt(0) drain per tile 391 us).  Here a fragment's 16 pixels x Cout channels go through
        // a wave-private LDS image and leave as whole rows -- 1 KB contiguous per store instruction: 198 us -- and the
        // GroupNorm statistics are summed from the registers BEFORE any store is issued (no drain, no store hazard: 234 us with
        // them; a fill of the same bytes takes 79 us, the general nine-step path 370).  Two channel passes for a third wave per
        // SIMD and the next tile's loads issued ahead of the stores were tried: 313 us, not kept.
        float st[MT][2];
        float4 bq[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            st[i][0] = 0.f; st[i][1] = 0.f;
            bq[i] = p.bias ? *(const float4*)(p.bias + i * 16 + fchunk * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        char* stage = sStage + wv * (16 * ROWB);
#pragma unroll
        for (int j = 0; j < PT; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float v0 = acc[i][j][0] + bq[i].x, v1 = acc[i][j][1] + bq[i].y, v2 = acc[i][j][2] + bq[i].z, v3 = acc[i][j][3] + bq[i].w;
                st[i][0] += (v0 + v1) + (v2 + v3);
                st[i][1] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                *(float4*)(stage + frow * ROWB + i * 64 + fchunk * 16) = make_float4(v0, v1, v2, v3);
            }
            __builtin_amdgcn_wave_barrier();
            char* out = (char*)p.y32 + ((size_t)m0 + j * 16) * (size_t)(MT * 64);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int lin = t * 1024 + lane * 16;          // byte offset inside the fragment's 16 x (MT * 64) output block
                const int row = lin / (MT * 64), col = lin - row * (MT * 64);
                const float4 v = *(const float4*)(stage + row * ROWB + col);
                *(float4*)(out + lin) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (p.gn_part != nullptr) {
            const int qpg = p.gn_cpg >> 2;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float v = st[i][k];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
                    if (qpg >= 2) v += __shfl_xor(v, 16, 64);
                    if (qpg == 4) v += __shfl_xor(v, 32, 64);
                    st[i][k] = v;
                }
            if (frow == 0 && (fchunk & (qpg - 1)) == 0) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int g = (i * 16 + fchunk * 4) / p.gn_cpg;
                    *(float2*)(p.gn_part + ((size_t)(m0 / 64) * 32 + g) * 2) = make_float2(st[i][0], st[i][1]);
                }
            }
        }
    }
}

// x f32 [B][H][W][ldx >= 3] (the dataloader's image), w_packed the general forward pack [9][Cout][8] (adap_pack_conv_weight of
// the 3-channel OIHW weight), Cout in {32, 64, 128} -> y32 / y16 [B][H][W][Cout]; honours adap_conv2d_next_gn_partial.
extern "C" int adap_conv3x3_rgb(const float* x_hwc, long ldx, const void* w_packed, const float* bias, float* y32, void* y16, int B,
                                int H, int W, int Cout, void* stream) {
    float* const gn_next = g_gn_next;
    const int gn_next_cpg = g_gn_next_cpg;
    g_gn_next = nullptr;
    g_gn_last_chunks = 0;
    ADAP_REQUIRE(x_hwc && w_packed && (y32 || y16), ADAP_ERR_SHAPE, "conv3x3_rgb: null pointer");
    ADAP_REQUIRE(B > 0 && H > 0 && W > 0 && ldx >= 3 && (Cout == 32 || Cout == 64 || Cout == 128), ADAP_ERR_UNSUPPORTED,
                 "conv3x3_rgb: B=%d H=%d W=%d Cout=%d", B, H, W, Cout);
    ADAP_REQUIRE(!y32 || ((uintptr_t)y32 % 16) == 0, ADAP_ERR_ALIGN, "conv3x3_rgb: y32 alignment");
    ADAP_REQUIRE(!y16 || ((uintptr_t)y16 % 8) == 0, ADAP_ERR_ALIGN, "conv3x3_rgb: y16 alignment");
    ADAP_REQUIRE(!bias || ((uintptr_t)bias % 16) == 0, ADAP_ERR_ALIGN, "conv3x3_rgb: bias alignment");
    const long M = (long)B * H * W;
    ADAP_REQUIRE(M < (1L << 31) && M * Cout * 4 < (1L << 32), ADAP_ERR_SHAPE, "conv3x3_rgb: output larger than 4 GiB");
    ConvParams p = {};
    p.x = x_hwc; p.ldx = ldx; p.w = (const uint16_t*)w_packed; p.bias = bias;
    p.y32 = y32; p.ldy32 = Cout; p.y16 = (uint16_t*)y16; p.ldy16 = Cout;
    p.B = B; p.Hin = H; p.Win = W; p.Cin = 8; p.Hout = H; p.Wout = W; p.Cout = Cout;
    p.KH = 3; p.KW = 3; p.stride = 1; p.pad = 1; p.alpha = 1.0f; p.ksplit = 1;
    if (gn_next && gn_next_cpg * 32 == Cout && gn_next_cpg % 4 == 0 && (H * W) % 256 == 0) {
        p.gn_part = gn_next;
        p.gn_cpg = gn_next_cpg;
        g_gn_last_chunks = H * W / 64;
    }
    const dim3 grid((unsigned)((M + 256 * RGB_TILES - 1) / (256 * RGB_TILES)));
    hipStream_t s = (hipStream_t)stream;
    if (Cout == 128) hipLaunchKernelGGL(conv3x3_rgb_kernel<8>, grid, dim3(256), 0, s, p);
    else if (Cout == 64) hipLaunchKernelGGL(conv3x3_rgb_kernel<4>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv3x3_rgb_kernel<2>, grid, dim3(256), 0, s, p);
    g_last_variant = 6000 + Cout;
    return adap_check_launch("conv3x3_rgb");
}

// ---------------------------------------------------------------------------------------------
// Weight packing: OIHW f32 (the checkpoint layout) -> bf16 [tap][Cout'][Cin'].
//   mode 0: forward     out[t][o][i]      = w[o][i][ky][kx],  t = ky*KW + kx   (Cin padded to cin_pad with zeros)
//   mode 1: data-grad   out[t][i][o]      = w[o][i][KH-1-ky][KW-1-kx]          (roles of Cin/Cout swapped, taps flipped)
// ---------------------------------------------------------------------------------------------
// out [taps][rows][cols] bf16 from w [O][I][KH][KW] f32.  mode 0: out[t][o][i] = w[o][i][t] (forward pack);
// mode 1: out[t][i][o] = w[o][i][taps-1-t] (data-gradient pack: roles swapped, taps flipped).  A workgroup stages a
// 32 (o) x 32 (i) x taps block through LDS so that both the f32 reads (runs of 32*taps floats per o) and the bf16 writes
// (runs of 32 along the packed layout's fast dim) are contiguous -- a training UNet repacks all 859.5 M weights after
// every optimiser step, and the element-wise gather this replaces ran at 1/7 of the HBM rate on the 3x3 layers.
#define PK_T 32
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int O, int I,
                                                          int taps, int mode, int rows, int cols) {
    extern __shared__ float pk_tile[];              // [PK_T o][PK_T i * taps (+1 pad)]
    const int pitch = PK_T * taps + 1;
    const int o0 = blockIdx.y * PK_T, i0 = blockIdx.x * PK_T;
    const int t = threadIdx.x;
    const int run = PK_T * taps;
    for (int idx = t; idx < PK_T * run; idx += 256) {
        const int oo = idx / run, e = idx - oo * run;
        const int ii = e / taps;
        float v = 0.f;
        if (o0 + oo < O && i0 + ii < I) v = w[((long)(o0 + oo) * I + i0) * taps + e];
        pk_tile[oo * pitch + e] = v;
    }
    __syncthreads();
    // mode 0: the output's (row, col) = (o, i); mode 1: (i, o)
    // two adjacent columns per thread: one 4-byte store (cols is even: a multiple of 8 or of 4)
    for (int idx = t; idx < taps * PK_T * (PK_T / 2); idx += 256) {
        const int cc = (idx % (PK_T / 2)) * 2;
        const int rr = (idx / (PK_T / 2)) % PK_T;
        const int tp = idx / (PK_T * (PK_T / 2));
        int r, c;
        float v0, v1;
        if (mode == 0) {
            r = o0 + rr; c = i0 + cc;
            v0 = pk_tile[rr * pitch + cc * taps + tp];
            v1 = pk_tile[rr * pitch + (cc + 1) * taps + tp];
        } else {
            r = i0 + rr; c = o0 + cc;
            v0 = pk_tile[cc * pitch + rr * taps + (taps - 1 - tp)];
            v1 = pk_tile[(cc + 1) * pitch + rr * taps + (taps - 1 - tp)];
        }
        if (r < rows && c < cols) *(uint32_t*)(out + ((long)tp * rows + r) * cols + c) = pack_bf16x2(v0, v1);
    }
}

extern "C" int adap_pack_conv_weight(const float* w_oihw, void* out_bf16, int O, int I, int KH, int KW,
                                     int mode, int rows, int cols, void* stream) {
    ADAP_REQUIRE(w_oihw && out_bf16, ADAP_ERR_SHAPE, "pack_weight: null pointer");
    ADAP_REQUIRE(mode == 0 || mode == 1, ADAP_ERR_UNSUPPORTED, "pack_weight: mode %d", mode);
    ADAP_REQUIRE(mode == 0 ? (rows >= O && cols >= I) : (rows >= I && cols >= O), ADAP_ERR_SHAPE,
                 "pack_weight: rows/cols too small");
    const int taps = KH * KW;
    ADAP_REQUIRE(taps >= 1 && taps <= 9, ADAP_ERR_UNSUPPORTED, "pack_weight: %d taps", taps);
    ADAP_REQUIRE(cols % 2 == 0 && ((uintptr_t)out_bf16 % 4) == 0, ADAP_ERR_ALIGN, "pack_weight: cols=%d must be even", cols);
    // tiles cover the PADDED extents (rows/cols beyond O/I are written as zeros)
    const int Op = mode == 0 ? rows : cols, Ip = mode == 0 ? cols : rows;
    dim3 grid((Ip + PK_T - 1) / PK_T, (Op + PK_T - 1) / PK_T);
    ADAP_REQUIRE(grid.y <= 65535, ADAP_ERR_SHAPE, "pack_weight: too many output channels");
    const size_t lds = (size_t)PK_T * (PK_T * taps + 1) * sizeof(float);
    hipLaunchKernelGGL(pack_weight_kernel, grid, dim3(256), lds, (hipStream_t)stream, w_oihw, (uint16_t*)out_bf16, O, I, taps,
                       mode, rows, cols);
    return adap_check_launch("pack_weight");
}


// ---------------------------------------------------------------------------------------------
// FeedForward's GEGLU fused into its two contractions (attention.py:32-59).  The 8C pre-activation is kept in the permuted
// channel order of ConvParams::epi (16 value channels, their 16 gates, the next 16 values, ...): w_packed / bias of
// adap_linear_geglu_fwd are the packs of ff.net.0.proj with their rows in that order.
// ---------------------------------------------------------------------------------------------
extern "C" int adap_linear_geglu_fwd(const void* x16, long ldx, const void* w_packed, const float* bias, void* h16, long ldh,
                                     void* out16, long ldo, long rows, int Cin, int C8, void* stream) {
    ADAP_REQUIRE(x16 && w_packed && h16 && out16, ADAP_ERR_SHAPE, "linear_geglu_fwd: null pointer");
    ADAP_REQUIRE(C8 % 32 == 0 && ldh >= C8 && ldo >= C8 / 2 && ldh % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)out16 % 8) == 0,
                 ADAP_ERR_ALIGN, "linear_geglu_fwd: C8=%d ldh=%ld ldo=%ld", C8, ldh, ldo);
    ADAP_REQUIRE(rows >= 1 && rows < (1L << 31), ADAP_ERR_SHAPE, "linear_geglu_fwd: rows");
    g_epi_ext = {1, (uint16_t*)out16, ldo, nullptr, 0};
    const int rc = adap_conv2d_nhwc(x16, 1, ldx, w_packed, bias, nullptr, 0, nullptr, 0, nullptr, 0, h16, ldh, 1, (int)rows, 1, Cin,
                                    (int)rows, 1, C8, 1, 1, 1, 0, 0, 1.0f, 1, nullptr, 1, 0, 0, 0, 0, stream);
    g_epi_ext = {0, nullptr, 0, nullptr, 0};
    return rc;
}

// d out [rows][C] (bf16, the gradient of ff.net.2's output) -> dh [rows][8C] (bf16, permuted order): ff.net.2's data gradient
// with d(a * gelu(gate)) applied in the epilogue; w_packed_bwd: ff.net.2's data-gradient pack [1][4C][C].
extern "C" int adap_linear_geglu_bwd(const void* g16, long ldg, const void* w_packed_bwd, const void* h16, long ldh, void* dh16,
                                     long lddh, long rows, int C, int C4, void* stream) {
    ADAP_REQUIRE(g16 && w_packed_bwd && h16 && dh16, ADAP_ERR_SHAPE, "linear_geglu_bwd: null pointer");
    ADAP_REQUIRE(C4 % 16 == 0 && ldh >= 2 * C4 && lddh >= 2 * C4 && ldh % 4 == 0 && lddh % 4 == 0 && ((uintptr_t)h16 % 8) == 0 &&
                 ((uintptr_t)dh16 % 8) == 0, ADAP_ERR_ALIGN, "linear_geglu_bwd: C4=%d ldh=%ld lddh=%ld", C4, ldh, lddh);
    ADAP_REQUIRE(rows >= 1 && rows < (1L << 31), ADAP_ERR_SHAPE, "linear_geglu_bwd: rows");
    g_epi_ext = {2, nullptr, 0, (const uint16_t*)h16, ldh};
    const int rc = adap_conv2d_nhwc(g16, 1, ldg, w_packed_bwd, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, dh16, lddh, 1, (int)rows, 1, C,
                                    (int)rows, 1, C4, 1, 1, 1, 0, 0, 1.0f, 1, nullptr, 1, 0, 0, 0, 0, stream);
    g_epi_ext = {0, nullptr, 0, nullptr, 0};
    return rc;
}
