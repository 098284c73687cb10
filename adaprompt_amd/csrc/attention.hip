// Fused (flash-style) multi-head attention for the SD-1.5 UNet on CDNA4 matrix cores.
//
// Reference semantics: CrossAttention.forward, attention.py:172-257 --
//   sim = (q k^T) * dim_head^-0.5 ; key mask -> -finfo.max ; softmax over keys ; out = attn v.
// dim_head is C/8 = 40 / 80 / 160 (not 64); self-attention has N = M up to 4096 tokens, cross
// attention M = 77 (or two 77-token halves for V and K in 'mix_hijk').  The reference
// materialises sim ([32,4096,4096] fp32 = 2.1 GB per layer at bs=4); here the score tile never
// leaves the registers.
//
// Tensors are token-major bf16: q [B,N,H*d], k,v [B,M,H*d] (outputs of the to_q/to_k/to_v
// contractions), out [B,N,H*d] bf16 (operand of to_out), lse [B,H,N] f32 for the backward.
//
// Structure (v_mfma_f32_32x32x16_bf16, fp32 softmax/accumulate): a workgroup = 4 waves = 128
// queries of one (batch, head); a wave owns 32 queries.  Scores are computed TRANSPOSED,
// S^T = K Q^T, so that the query is on the lane and the keys are in the accumulator registers:
// row max / row sum are in-lane plus one half-wave exchange, the running rescale is a per-lane
// scalar, and the accumulator is already the B operand of O^T = V^T P^T (guide section 3, "An
// accumulator tile as the next MFMA's operand").  K tiles are read by rows (ds_read_b128), V by
// the hardware-transposing ds_read_b64_tr_b16 from the same row-major [key][d] image, so no
// transposed copy of V exists anywhere.
//
// Backward = two kernels without atomics (bit-reproducible): dQ is query-stationary (same loop
// as the forward), dK/dV are key-stationary (a wave owns 32 keys and sweeps the queries); P is
// recomputed from the saved LSE.
#include <algorithm>
#include <type_traits>
#include "common.h"
#include <float.h>

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
}

// A-operand fragment of X^T for a [row][col] bf16 LDS image X: element j of lane (c = lane&31, h = lane>>5)
// is X[row0 + 8*(j>>2) + 4h + (j&3)][col0 + c]  (the k-order of an accumulator used as B operand).
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* img, int stride, int row0, int col0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const char* p = img + (row0 + 4 * (g >> 1) + (i >> 2)) * stride + (col0 + 16 * (g & 1) + 4 * (i & 3)) * 2;
    bf16x4 lo = lds_tr16(p);
    bf16x4 hi = lds_tr16(p + 8 * stride);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int sh) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * sh + j];
    return r;
}

// A per-query (per-row) f32 addend of the scores that rides through the matrix core instead of the vector unit: with a spare
// K-dimension chunk (dim_head = 40 pads to 48) the key side carries 1.0 in the chunk's first columns and the query side the
// addend, split into bf16 terms -- hi + mid + lo reproduce an f32 to 2^-24 of its size, and the products 1.0 * term are exact.
// (The accumulator then starts from the inline constant 0: no 16-register copy of the addend per score tile.)
__device__ __forceinline__ uint2 bf16_split3(float x) {
    const float hi = bf16_to_f32(f32_to_bf16(x));
    const float r1 = x - hi;
    const float mid = bf16_to_f32(f32_to_bf16(r1));
    const float lo = r1 - mid;
    return make_uint2((uint32_t)f32_to_bf16(hi) | ((uint32_t)f32_to_bf16(mid) << 16), (uint32_t)f32_to_bf16(lo));
}
#define BF16_ONES3_X 0x3F803F80u      // (1.0, 1.0) and
#define BF16_ONES3_Y 0x00003F80u      // (1.0, 0): the key side of a three-term addend

#define TOK_MAXG 4
struct AttnParams {
    const uint16_t* q; long ldq;
    const uint16_t* k; long ldk;
    const uint16_t* v; long ldv;
    const uint8_t* kmask;       // [B][M] (1 = keep) or null
    const int* mcount;          // [B] or null: sample b attends to its first mcount[b] keys only (compacted keys, rows >= count unused)
    uint16_t* o; long ldo;      // fwd out (bf16)
    float* lse;                 // [B][H][N]
    // backward
    const uint16_t* dout; long lddo;   // bf16 [B,N,H*d]
    float* delta;                      // [B][H][N]: written by the dQ kernel, read by the dK/dV kernel
    float* dq32; uint16_t* dq16; long lddq;
    float* dk32; uint16_t* dk16; long lddk;
    float* dv32; uint16_t* dv16; long lddv;
    int B, H, N, M, d;
    float scale;
    int pre;                    // q already carries scale * log2(e) (folded into the projection's weight pack): the scores come out
                                // of the matrix core in the exp2 domain and, with the reference point / -lse as the accumulator's
                                // initial value, go into v_exp_f32 as they are -- one VALU instruction per score less
    int qsplit;                 // dK/dV kernel: the query range is cut into qsplit slices (blockIdx.z), f32 partials
    int xcd;                    // bit 0 / 1 / 2 (forward / dQ / dK,dV kernel): the workgroups of one (batch, head) -- and of
                                // neighbouring heads -- share an XCD (attn_wg_coords)
    float* part;                // [qsplit][2][B*M][H*d] when qsplit > 1
    // gradient of the cross-attention token maps folded into this backward (adap_attention_bwd_tok): dq += scale * dT . kw,
    // dk += scale * w . gq, with kw = w^T K and gq = dT^T Q from adap_attention_tokmap_prep
    const float* tok_dt;        // [B][H][N][G] or null
    const float* tok_w;         // [B][M][G]
    const float* tok_kw;        // [B*H][G][d]
    const float* tok_gq;        // [B*H][G][d]
    int tok_G;
    // ping-pong forward only
    int prio_mode;              // which phase runs at raised priority: 1 = matrix (default), 2 = vector, 0 = neither
    unsigned long long* stamps; // diagnostic s_memtime stamps (tools/attn_stamps.py) or null
};

// sample b attends to its first key_count[b] keys, clamped to [1, M]: a count <= 0 would leave the softmax without a
// denominator (l = 0 -> NaN output); the C ABI documents key_count >= 1 and treats anything smaller as 1
__device__ __forceinline__ int attn_key_count(const AttnParams& p, int b) {
    return p.mcount ? max(1, min(p.mcount[b], p.M)) : p.M;
}

// Which (block, batch * head, slice) a workgroup works on.  The grids are (blocks of one head, B * H, slices); the hardware deals
// workgroups round-robin over the 8 XCDs in launch order, so with the plain blockIdx map the 16-32 blocks of one (batch, head)
// land on ALL eight XCDs and every XCD's 4 MiB L2 fetches the K / V (dK/dV kernel: Q / dO) of all heads: 2.3x (forward) and
// 5.2x (dK/dV) the algorithmic bytes at N 4096 (profiles/r04_pmc_traffic.json).  Remapped, workgroup L -> XCD group L % 8,
// slot L / 8, and an XCD group walks B*H / 8 CONSECUTIVE (batch, head) rows, all blocks of a row before the next: the operands a
// row's workgroups share are fetched into one L2, and rows that run side by side on an XCD are neighbouring heads of one sample
// -- their 80-byte head slices of a token's row share 128-byte lines (a head alone touches 1.5 lines per 80 bytes).
// Placement only changes speed; B*H not a multiple of 8 keeps the plain map.
__device__ __forceinline__ void attn_wg_coords(const AttnParams& p, int which, int& bx, int& bh, int& bz) {
    const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    if (!(p.xcd & which) || (gy & 7u)) { bx = blockIdx.x; bh = blockIdx.y; bz = blockIdx.z; return; }
    const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned grp = L & 7u, slot = L >> 3, per = gx * gz;
    const unsigned row = slot / per, r = slot - row * per;
    bh = (int)(grp * (gy >> 3) + row);
    bz = (int)(r / gx);
    bx = (int)(r - (unsigned)bz * gx);
}

#define MAX_SLACK 5.0f          // see attn_fwd_kernel's running reference point

// max over the two lanes l and l ^ 32 -- the two halves of a score column in the 32x32 accumulator layout -- without the LDS
// round trip of __shfl_xor (ds_bpermute + lgkmcnt wait, queued behind the partner wave's fragment reads): gfx950's
// v_permlane32_swap exchanges one register's upper 32 lanes with the other's lower 32 in the vector pipe.
__device__ __forceinline__ float xor32_max(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}

template <int KS>
struct TileGeom {
    static constexpr int DPK = KS * 16;
    static constexpr int NCH = DPK / 8;                 // 16-byte chunks per padded row
    static constexpr int RSTRIDE = ((NCH | 1)) * 16;    // odd chunk count: conflict-free ds_read_b128 by rows
};

template <int VT>
struct VGeom {
    static constexpr int VSTRIDE = (VT % 2 == 1) ? 64 * VT : 64 * VT + 64;   // == 64 (mod 128): conflict-free tr reads
};

// stage a [rows x d] bf16 tile (row-major in HBM with leading dim ld) into LDS with the given stride, zero padded to
// nch chunks per row; rows >= nvalid are zero.
__device__ __forceinline__ void stage_tile(char* dst, int stride, int nch, const uint16_t* src, long ld, int nrows,
                                           int nvalid, int d, int tid) {
    for (int idx = tid; idx < nrows * nch; idx += 256) {
        int row = idx / nch, ch = idx - row * nch;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < nvalid && ch * 8 < d) v = *(const uint4*)(src + (size_t)row * ld + ch * 8);
        *(uint4*)(dst + row * stride + ch * 16) = v;
    }
}

// Register-staged prefetch of a [64 x NCH*8] bf16 tile: the global loads of tile t+1 are issued before tile t is
// multiplied and written to LDS after it (guide T14), so HBM/L2 latency hides under the MFMA phase.  The chunk ->
// (row, column) decode is done once per kernel (TileMap), not per tile.
template <int NCH, int NT = 256>
struct TileRegs {
    static constexpr int CPT = (64 * NCH + NT - 1) / NT;
    uint4 v[CPT];
};

template <int NCH, int NT = 256>
struct TileMap {
    static constexpr int CPT = (64 * NCH + NT - 1) / NT;
    int row[CPT];       // tile row of chunk i, or 64 (never valid)
    int goff[CPT];      // element offset in the global tile
    int loff[CPT];      // byte offset in the LDS image, or -1
    bool one[CPT];      // this chunk starts at the "ones" column (V tile: its first element is bf16 1.0, the rest 0)
    // skip_chunk: this chunk of every row is written by somebody else (the dK/dV kernel's per-row addends), not by the tile's store
    __device__ __forceinline__ void init(long ld, int stride, int d, int tid, int ones_chunk = -1, int skip_chunk = -1) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            int idx = tid + NT * i;
            int r = idx / NCH, ch = idx - r * NCH;
            bool in = idx < 64 * NCH;
            row[i] = (in && ch * 8 < d) ? r : 64;
            goff[i] = (int)(r * ld) + ch * 8;
            loff[i] = (in && ch != skip_chunk) ? r * stride + ch * 16 : -1;
            one[i] = in && ch == ones_chunk;
        }
    }
};

template <int NCH>
__device__ __forceinline__ void tile_load(TileRegs<NCH>& r, const TileMap<NCH>& mp, const uint16_t* src, int nvalid) {
#pragma unroll
    for (int i = 0; i < TileRegs<NCH>::CPT; ++i) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (mp.row[i] < nvalid) v = *(const uint4*)(src + mp.goff[i]);
        if (mp.one[i]) v.x = 0x3F80u;
        r.v[i] = v;
    }
}

// The same for a staging pipeline more than one step deep.  The load has no divergent branch around it (an invalid chunk reads
// the tile's first chunk), so the number of loads in flight is the same on every path and the compiler can wait for an OLDER
// set with a counted vmcnt; zero fill and the ones column are applied when the registers are STORED (tile_store_fix, same
// nvalid) -- applied at load time they would be the loaded registers' first use and drain the queue in the step that issued them.
template <int NCH, int NT>
__device__ __forceinline__ void tile_load_raw(TileRegs<NCH, NT>& r, const TileMap<NCH, NT>& mp, const uint16_t* src, int nvalid) {
#pragma unroll
    for (int i = 0; i < TileRegs<NCH, NT>::CPT; ++i) r.v[i] = *(const uint4*)(src + (mp.row[i] < nvalid ? mp.goff[i] : 0));
}

template <int NCH, int NT>
__device__ __forceinline__ void tile_store_fix(const TileRegs<NCH, NT>& r, const TileMap<NCH, NT>& mp, char* dst, int nvalid) {
#pragma unroll
    for (int i = 0; i < TileRegs<NCH, NT>::CPT; ++i) {
        uint4 v = r.v[i];
        if (!(mp.row[i] < nvalid)) v = make_uint4(0, 0, 0, 0);
        if (mp.one[i]) v.x = 0x3F80u;
        if (mp.loff[i] >= 0) *(uint4*)(dst + mp.loff[i]) = v;
    }
}

template <int NCH>
__device__ __forceinline__ void tile_store(const TileRegs<NCH>& r, const TileMap<NCH>& mp, char* dst) {
#pragma unroll
    for (int i = 0; i < TileRegs<NCH>::CPT; ++i)
        if (mp.loff[i] >= 0) *(uint4*)(dst + mp.loff[i]) = r.v[i];
}

// =============================================================================================
// forward
// =============================================================================================
// QB = 32-row query blocks per wave.  QB = 2 (long sequences): every K fragment and every transposed V fragment read
// from LDS feeds two MFMAs instead of one, and the two blocks' softmax chains are independent work inside one wave.
// PRE: the pre-scaled-query form (AttnParams::pre) -- a template parameter, not a run-time branch: with the branch in the loop
// the default form lost its schedule (B4 h8 N4096 d40: 152 -> 284 us; MFMA pipe 27 -> 17 %)
template <int KS, int VT, int QB, bool PRE>
__global__ __launch_bounds__(256, (QB == 1 && KS <= 6) || (QB == 2 && KS <= 4) ? 2 : 1) void attn_fwd_kernel(AttnParams p) {
    using G = TileGeom<KS>;
    constexpr int KSTRIDE = G::RSTRIDE;
    constexpr int VSTRIDE = VGeom<VT>::VSTRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // K / V tiles and the key bias are double-buffered: tile kt+1 is written (from the registers its global loads
    // landed in) while the other waves may still be multiplying tile kt, so the loop needs ONE barrier per tile
    constexpr int TILE = 64 * KSTRIDE + 64 * VSTRIDE;
    char* sKV = smem;                                   // [2][ K [64][KSTRIDE] | V [64][VSTRIDE] ]
    float* sBiasAll = (float*)(smem + 2 * TILE);        // [2][68]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    int bx, bh, bz_;
    attn_wg_coords(p, 1, bx, bh, bz_);
    const int b = bh / p.H, head = bh - b * p.H;
    const int q0 = bx * (128 * QB) + wave * (32 * QB) + c;      // query of block qb: q0 + 32 * qb
    const int d = p.d;
    const float cs = p.scale * 1.4426950408889634f;
    constexpr bool pre = PRE;

    // Q fragments stay exactly the caller's bf16 values: scale*log2(e) is applied in f32 inside the exponent's fma
    // (pre-multiplying Q would round q*cs to bf16 again and cost ~2e-4 of LSE accuracy for ~1% of a step)
    bf16x8 qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            int ch = 2 * s + h, q = q0 + 32 * qb;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < p.N && ch * 8 < d) v = *(const uint4*)(p.q + ((size_t)b * p.N + q) * p.ldq + head * d + ch * 8);
            qf[qb][s] = __builtin_bit_cast(bf16x8, v);
        }

    f32x16 O[QB][VT];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[qb][vt][r] = 0.f;
    float m[QB], l[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { m[qb] = -INFINITY; l[qb] = 0.f; }
    // When the V tile has a spare column (32*VT > d) it is set to 1.0, so row d of O^T accumulates sum_k p -- the
    // softmax denominator comes out of the PV MFMA for free (d = 40, 80; not 160).  The 1.0 is planted by the tile
    // staging itself (TileMap::one), not by a separate write.
    const bool ones_col = 32 * VT > d && (d >> 3) < G::NCH;

    const uint16_t* kb = p.k + (size_t)b * p.M * p.ldk + head * d;
    const uint16_t* vb = p.v + (size_t)b * p.M * p.ldv + head * d;
    const int Mb = attn_key_count(p, b);          // this sample's key count
    const int ntiles = (Mb + 63) / 64;
    TileRegs<G::NCH> rK, rV;
    TileMap<G::NCH> mapK, mapV;
    // pre-scaled queries with a spare K chunk: the reference point -m rides in through that chunk (the K tile carries 1.0 in
    // column d, the query fragment -m there) -- m is kept bf16-representable for it, which is as good a reference point as any
    const bool padm = pre && (d >> 3) < G::NCH && !(p.xcd & 16);
    mapK.init(p.ldk, KSTRIDE, d, tid, padm ? (d >> 3) : -1);
    mapV.init(p.ldv, VSTRIDE, d, tid, ones_col ? (d >> 3) : -1);
    const int pad_s = d >> 4;                 // the spare chunk is chunk d/8: fragment slice (d/8)/2 of the lanes h == (d/8)&1
    const bool pad_lane = padm && h == ((d >> 3) & 1);
    // per-key additive bias of a tile (ragged last tile: -inf, masked keys: -FLT_MAX) and whether the tile has any
    auto key_bias = [&](int key0, float* sBias) {
        if (tid < 64) {
            int key = key0 + tid;
            float bias = 0.f;
            if (key >= Mb) bias = -INFINITY;
            else if (p.kmask && !p.kmask[(size_t)b * p.M + key]) bias = -FLT_MAX;
            sBias[tid] = bias;
            unsigned long long any = __ballot(bias != 0.f);
            if (tid == 0) sBias[64] = any ? 1.f : 0.f;
        }
    };
    tile_load(rK, mapK, kb, min(64, Mb));
    tile_load(rV, mapV, vb, min(64, Mb));
    tile_store(rK, mapK, sKV);
    tile_store(rV, mapV, sKV + 64 * KSTRIDE);
    key_bias(0, sBiasAll);
    __syncthreads();
    // the tile loop in two compiled forms: PADM = the reference point rides in through the spare K chunk (the score tiles then
    // start from the inline constant 0; with a run-time flag in one loop body they start from a register, 16 copies per tile)
    auto tiles = [&](auto padc) {
        constexpr bool PADM = decltype(padc)::value;
        for (int kt = 0; kt < ntiles; ++kt) {
            const bool more = kt + 1 < ntiles;
            const char* sK = sKV + (kt & 1) * TILE;
            const char* sV = sK + 64 * KSTRIDE;
            const float* sBias = sBiasAll + (kt & 1) * 68;
            if (more) {
                const int key1 = (kt + 1) * 64;
                tile_load(rK, mapK, kb + (size_t)key1 * p.ldk, min(64, Mb - key1));
                tile_load(rV, mapV, vb + (size_t)key1 * p.ldv, min(64, Mb - key1));
            }
            f32x16 S[QB][2];
            float sinit[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) sinit[qb] = (pre && !PADM && m[qb] != -INFINITY) ? -m[qb] : 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) S[qb][t][r] = sinit[qb];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    bf16x8 kf = *(const bf16x8*)(sK + (32 * t + c) * KSTRIDE + (2 * s + h) * 16);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        S[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][s], S[qb][t], 0, 0, 0);
                }
            }
            // wave-uniform: ragged / masked tile (not even looked at when there is no key mask and the tile is not the ragged last one)
            const bool biased = (p.kmask != nullptr || (!more && (Mb & 63) != 0)) && sBias[64] != 0.f;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (biased) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float4 bv = *(const float4*)(sBias + 32 * t + 8 * g + 4 * h);
                            S[qb][t][4 * g] += bv.x; S[qb][t][4 * g + 1] += bv.y;
                            S[qb][t][4 * g + 2] += bv.z; S[qb][t][4 * g + 3] += bv.w;
                        }
                }
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[qb][t][r]);
                mx = xor32_max(mx);
                // running reference point in the exp2 domain (cs > 0): raised only when a row's max outgrows it by more than
                // 2^MAX_SLACK (or it is still -inf) -- p <= 2^MAX_SLACK keeps its full f32 / bf16 relative precision, and the O
                // rescale below (64 multiplies) then runs on a handful of tiles instead of on every second one
                if constexpr (pre) {
                    // S is already in the exp2 domain and relative to the reference point m (absolute while m is still -inf)
                    const bool first = m[qb] == -INFINITY;
                    const bool grow = first ? (mx > -INFINITY) : (mx > MAX_SLACK);
                    if (__any(grow)) {                      // wave-uniform, rare: the reference moves up by `shift`
                        float mnew = first ? mx : m[qb] + mx;
                        if (PADM) mnew = bf16_to_f32(f32_to_bf16(mnew));   // (it travels in a Q fragment: a bf16 value)
                        const float shift = !grow ? 0.f : (first ? mnew : mnew - m[qb]);
                        if (PADM && grow) {
#pragma unroll
                            for (int s2 = 0; s2 < KS; ++s2)
                                if (s2 == pad_s && pad_lane) {
                                    uint4 qv = __builtin_bit_cast(uint4, qf[qb][s2]);
                                    qv.x = (uint32_t)f32_to_bf16(-mnew);
                                    qf[qb][s2] = __builtin_bit_cast(bf16x8, qv);
                                }
                        }
                        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-shift);
#pragma unroll
                        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) O[qb][vt][r] *= alpha;
                        l[qb] *= alpha;
                        if (grow) m[qb] = mnew;
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) S[qb][t][r] -= shift;
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) S[qb][t][r] = __builtin_amdgcn_exp2f(S[qb][t][r]);
                } else {
                    const float mcand = mx * cs;
                    const bool grow = mcand > m[qb] + MAX_SLACK;
                    if (__any(grow)) {                          // wave-uniform
                        const float mnew = grow ? mcand : m[qb];
                        const float alpha = __builtin_amdgcn_exp2f(m[qb] - mnew);      // 1 for the rows that keep their reference
#pragma unroll
                        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) O[qb][vt][r] *= alpha;
                        l[qb] *= alpha;
                        m[qb] = mnew;
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) S[qb][t][r] = __builtin_amdgcn_exp2f(fmaf(S[qb][t][r], cs, -m[qb]));
                }
                if (!ones_col) {
                    float psum = 0.f;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) psum += S[qb][t][r];
                    l[qb] += psum;
                }
            }
            // O^T += V^T P^T
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    bf16x8 pf[QB];
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) pf[qb] = acc_to_frag(S[qb][t], sh);
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) {
                        bf16x8 vf = lds_tr_frag(sV, VSTRIDE, 32 * t + 16 * sh, 32 * vt, lane);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb)
                            O[qb][vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb], O[qb][vt], 0, 0, 0);
                    }
                }
            if (more) {                                     // the other buffer: its last readers passed the previous barrier
                char* nK = sKV + ((kt + 1) & 1) * TILE;
                tile_store(rK, mapK, nK);
                tile_store(rV, mapV, nK + 64 * KSTRIDE);
                key_bias((kt + 1) * 64, sBiasAll + ((kt + 1) & 1) * 68);
            }
            __syncthreads();
        }
    };
    if constexpr (pre) {
        if (padm) tiles(std::true_type{});
        else tiles(std::false_type{});
    } else {
        tiles(std::false_type{});
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q0 + 32 * qb;
        float ltot;
        if (ones_col) {
            // row d of O^T lives in tile d/32, register (rin&3) + 4*(rin>>3) of the lane half (rin>>2)&1, rin = d%32
            const int vt0 = d >> 5, rin = d & 31, hh = (rin >> 2) & 1, reg = (rin & 3) + 4 * (rin >> 3);
            float lv = 0.f;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (vt == vt0 && r == reg) lv = O[qb][vt][r];
            ltot = __shfl(lv, c + 32 * hh, 64);
        } else {
            ltot = l[qb] + __shfl_xor(l[qb], 32, 64);
        }
        const float inv = 1.0f / ltot;
        if (q < p.N) {
            uint16_t* orow = p.o + ((size_t)b * p.N + q) * p.ldo + head * d;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int d0 = 32 * vt + 8 * g + 4 * h;
                    if (d0 < d) {
                        uint2 w;
                        w.x = pack_bf16x2(O[qb][vt][4 * g] * inv, O[qb][vt][4 * g + 1] * inv);
                        w.y = pack_bf16x2(O[qb][vt][4 * g + 2] * inv, O[qb][vt][4 * g + 3] * inv);
                        *(uint2*)(orow + d0) = w;
                    }
                }
            if (h == 0 && p.lse) p.lse[((size_t)b * p.H + head) * p.N + q] = (m[qb] + log2f(ltot)) * 0.6931471805599453f;
        }
    }
}

// =============================================================================================
// forward, ping-pong form (long sequences, short heads: the 64x64 level, N = 4096, d = 40)
// =============================================================================================
// At d = 40 a 64 x 64 score tile costs a wave 28 MFMAs (896 matrix-pipe cycles) and ~1100 cycles of softmax VALU issue
// (66 v_exp at 8 cycles, 64 fma, 32 max3, 32 cvt): neither pipe can hide behind the other inside ONE wave's dependent
// stream, and in the kernel above the two waves of a SIMD (different workgroups) drift through the same phases together
// (MFMA pipe 27 % busy).  Here a workgroup is 8 waves = 512 queries, waves w and w + 4 share a SIMD, and the two halves run
// HALF A TILE APART across one barrier per phase: while waves 0-3 exponentiate tile t (VALU), waves 4-7 multiply
// (P V of tile t-1, then Q K^T of tile t: 28 back-to-back MFMAs), then they swap.  Every SIMD always has one wave in its
// matrix phase and one in its vector phase.  K tiles are double-, V tiles and the key bias triple-buffered in LDS, staged
// through registers one tile ahead by all 512 threads; the running max is only raised when it would grow by more than
// 2^PP_THR (guide T13: the rescale of O then almost never runs; p <= 2^PP_THR stays exact in the f32 exponent path and
// bf16 keeps its relative precision), which keeps the vector phase branch-free in the common case.
#define PP_THR 5.0f

template <int KS, int VT>
__global__ __launch_bounds__(512, 1) void attn_fwd_pp_kernel(AttnParams p) {
    using G = TileGeom<KS>;
    constexpr int KSTRIDE = G::RSTRIDE;
    constexpr int VSTRIDE = VGeom<VT>::VSTRIDE;
    constexpr int NCH = G::NCH;
    constexpr int KT = 64 * KSTRIDE, VTB = 64 * VSTRIDE;
    static_assert(64 * NCH <= 512, "one 16-byte chunk per thread and tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                                    // [2][64][KSTRIDE]
    char* sV = smem + 2 * KT;                           // [3][64][VSTRIDE]
    float* sBiasAll = (float*)(smem + 2 * KT + 3 * VTB);  // [3][68]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2;                          // 0: waves 0-3, 1: waves 4-7 (the SIMD partners), half a tile behind
    const int c = lane & 31, h = lane >> 5;
    int bx, bh, bz_;
    attn_wg_coords(p, 1, bx, bh, bz_);
    const int b = bh / p.H, head = bh - b * p.H;
    const int q0 = bx * 512 + wave * 64 + c;    // query of block qb: q0 + 32 * qb
    const int d = p.d;
    const float cs = p.scale * 1.4426950408889634f;

    bf16x8 qf[2][KS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            int ch = 2 * s + h, q = q0 + 32 * qb;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < p.N && ch * 8 < d) v = *(const uint4*)(p.q + ((size_t)b * p.N + q) * p.ldq + head * d + ch * 8);
            qf[qb][s] = __builtin_bit_cast(bf16x8, v);
        }
    f32x16 O[2][VT];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[qb][vt][r] = 0.f;
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    const bool ones_col = 32 * VT > d && (d >> 3) < NCH;

    const uint16_t* kb = p.k + (size_t)b * p.M * p.ldk + head * d;
    const uint16_t* vb = p.v + (size_t)b * p.M * p.ldv + head * d;
    const int Mb = attn_key_count(p, b);          // this sample's key count
    const int ntiles = (Mb + 63) / 64;
    // staging map: thread -> chunk (row, 16-byte column) of a 64-row tile
    const int srow = tid / NCH, sch = tid - srow * NCH;
    const bool sin = tid < 64 * NCH && sch * 8 < d;       // chunk holds data (else zero padding)
    const bool slive = tid < 64 * NCH;
    const bool sone = slive && ones_col && sch == (d >> 3);
    uint4 rK = make_uint4(0, 0, 0, 0), rV = make_uint4(0, 0, 0, 0);
    auto stage_load = [&](int s) {
        const int key0 = s * 64, nvalid = min(64, Mb - key0);
        rK = make_uint4(0, 0, 0, 0);
        rV = make_uint4(0, 0, 0, 0);
        if (sin && srow < nvalid) {
            rK = *(const uint4*)(kb + (size_t)(key0 + srow) * p.ldk + sch * 8);
            rV = *(const uint4*)(vb + (size_t)(key0 + srow) * p.ldv + sch * 8);
        }
        if (sone) rV.x = 0x3F80u;
    };
    auto stage_store = [&](int s) {
        if (slive) {
            *(uint4*)(sK + (s & 1) * KT + srow * KSTRIDE + sch * 16) = rK;
            *(uint4*)(sV + (s % 3) * VTB + srow * VSTRIDE + sch * 16) = rV;
        }
        if (tid < 64) {
            float* sBias = sBiasAll + (s % 3) * 68;
            const int key = s * 64 + tid;
            float bias = 0.f;
            if (key >= Mb) bias = -INFINITY;
            else if (p.kmask && !p.kmask[(size_t)b * p.M + key]) bias = -FLT_MAX;
            sBias[tid] = bias;
            unsigned long long any = __ballot(bias != 0.f);
            if (tid == 0) sBias[64] = any ? 1.f : 0.f;
        }
    };
    f32x16 S[2][2];
    bf16x8 pf[2][2][2];
    // The matrix phase as ONE software-pipelined stream of fragment groups: the LDS reads of group i + 1 are issued BEFORE
    // the MFMAs of group i, so a read's latency hides under 2-4 MFMAs instead of stalling the pipe in front of every group
    // (left to itself hipcc sinks each ds_read to just above its first use: 14 exposed LDS round trips per tile).
    //   groups 0..3  : P V of tile t, (tt, sh) = (g >> 1, g & 1): VT transposed V fragments, 2 * VT MFMAs
    //   groups 4..4+2*KS-1 : Q K^T of tile t + 1, (tt, s) = ((g-4) / KS, (g-4) % KS): one K fragment, 2 MFMAs
    auto v_frag = [&](const char* vt_, int g, int vt) { return lds_tr_frag(vt_, VSTRIDE, 32 * (g >> 1) + 16 * (g & 1), 32 * vt, lane); };
    auto k_frag = [&](const char* kt, int g) {
        const int tt = g / KS, s = g - tt * KS;
        return *(const bf16x8*)(kt + (32 * tt + c) * KSTRIDE + (2 * s + h) * 16);
    };
    auto mm_phase = [&](int t, bool do_pv, bool do_qk) {
        const char* vt_ = sV + (t % 3) * VTB;
        const char* kt = sK + ((t + 1) & 1) * KT;
        bf16x8 vcur[VT], vnext[VT], kcur, knext;
        if (do_pv) {
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) vcur[vt] = v_frag(vt_, 0, vt);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) {
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) vnext[vt] = v_frag(vt_, g + 1, vt);
                } else if (do_qk) {
                    knext = k_frag(kt, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
                        O[qb][vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vcur[vt], pf[qb][g >> 1][g & 1], O[qb][vt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) vcur[vt] = vnext[vt];
            }
        } else if (do_qk) {
            knext = k_frag(kt, 0);
        }
        if (do_qk) {
#pragma unroll
            for (int g = 0; g < 2 * KS; ++g) {
                kcur = knext;
                if (g + 1 < 2 * KS) knext = k_frag(kt, g + 1);
                __builtin_amdgcn_sched_barrier(0);
                const int tt = g / KS, s = g - tt * KS;
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    if (s == 0) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        S[qb][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kcur, qf[qb][s], z, 0, 0, 0);
                    } else {
                        S[qb][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kcur, qf[qb][s], S[qb][tt], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto softmax = [&](int t) {                             // S -> P (bf16 fragments), running max with a threshold
        const float* sBias = sBiasAll + (t % 3) * 68;
        // wave-uniform: ragged / masked tile (the flag sits in LDS; not even looked at when neither can be the case)
        const bool biased = (p.kmask != nullptr || ((Mb & 63) != 0 && t == ntiles - 1)) && sBias[64] != 0.f;
        if (biased) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float4 bv = *(const float4*)(sBias + 32 * tt + 8 * g + 4 * h);
                        S[qb][tt][4 * g] += bv.x; S[qb][tt][4 * g + 1] += bv.y;
                        S[qb][tt][4 * g + 2] += bv.z; S[qb][tt][4 * g + 3] += bv.w;
                    }
        }
        // both query blocks in ONE straight-line region (their chains interleave), one rare branch for the rescale
        float mcand[2];
        bool grow[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[qb][tt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            mcand[qb] = mx * cs;                            // exp2 domain (cs > 0)
            // raise the reference point only when the row's max outgrows it by more than 2^PP_THR (or it is still -inf)
            grow[qb] = mcand[qb] > m[qb] + PP_THR;
        }
        if (__any(grow[0] || grow[1])) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float mnew = grow[qb] ? mcand[qb] : m[qb];
                const float alpha = __builtin_amdgcn_exp2f(m[qb] - mnew);      // 1 for rows that keep their reference
#pragma unroll
                for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[qb][vt][r] *= alpha;
                l[qb] *= alpha;
                m[qb] = mnew;
            }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[qb][tt][r] = fmaf(S[qb][tt][r], cs, -m[qb]);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[qb][tt][r] = __builtin_amdgcn_exp2f(S[qb][tt][r]);
        if (!ones_col) {                                    // (d = 40, 80: the denominator comes out of the P V MFMA)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float psum = 0.f;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) psum += S[qb][tt][r];
                l[qb] += psum;
            }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) pf[qb][tt][sh] = acc_to_frag(S[qb][tt], sh);
    };

    // ---- prologue: tiles 0 and 1 in LDS, S(0) in registers
    stage_load(0);
    stage_store(0);
    if (ntiles > 1) {
        stage_load(1);
        stage_store(1);
    }
    __syncthreads();
    mm_phase(-1, false, true);                              // S = K(0) Q^T
    // ---- phases: group g exponentiates tile t in phase 2t + g and multiplies (P V of t, Q K^T of t + 1) in phase 2t + 1 + g
    const int nphases = 2 * ntiles + 1;
    // diagnostic (tools/attn_stamps.py): workgroup (0,0)'s waves 0 and 4 stamp every phase: entry, work done, barrier passed
    unsigned long long* stamps = (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave & 3) == 0)
                                     ? p.stamps : nullptr;
    for (int ph = 0; ph < nphases; ++ph) {
        if (stamps) stamps[(ph * 2 + grp) * 3 + 0] = __builtin_amdgcn_s_memtime();
        if ((ph & 1) == 0) {                                // tile ph/2 + 2 starts its trip to LDS
            const int s = (ph >> 1) + 2;
            if (s < ntiles) stage_load(s);
        }
        const int r = ph - grp;
        if (r >= 0 && (r >> 1) < ntiles) {
            const int t = r >> 1;
            if ((r & 1) == 0) {
                if (p.prio_mode == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
                softmax(t);
                __builtin_amdgcn_s_setprio(0);
            } else {
                // the matrix phase issues first: its 28 MFMAs take 8 issue cycles each and keep the pipe busy for 32; left at equal
                // priority the vector-phase partner (when it is the older wave) starves them (stamps: 2340 vs 1590 ticks)
                if (p.prio_mode == 1) __builtin_amdgcn_s_setprio(2);
                mm_phase(t, true, t + 1 < ntiles);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        if (ph & 1) {                                       // ... and lands: its buffers' last readers passed the previous barrier
            const int s = (ph + 3) >> 1;
            if (s < ntiles) stage_store(s);
        }
        if (stamps) stamps[(ph * 2 + grp) * 3 + 1] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (stamps) stamps[(ph * 2 + grp) * 3 + 2] = __builtin_amdgcn_s_memtime();
    }
    // ---- epilogue (as in attn_fwd_kernel)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + 32 * qb;
        float ltot;
        if (ones_col) {
            const int vt0 = d >> 5, rin = d & 31, hh = (rin >> 2) & 1, reg = (rin & 3) + 4 * (rin >> 3);
            float lv = 0.f;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (vt == vt0 && r == reg) lv = O[qb][vt][r];
            ltot = __shfl(lv, c + 32 * hh, 64);
        } else {
            ltot = l[qb] + __shfl_xor(l[qb], 32, 64);
        }
        const float inv = 1.0f / ltot;
        if (q < p.N) {
            uint16_t* orow = p.o + ((size_t)b * p.N + q) * p.ldo + head * d;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int d0 = 32 * vt + 8 * g + 4 * h;
                    if (d0 < d) {
                        uint2 w;
                        w.x = pack_bf16x2(O[qb][vt][4 * g] * inv, O[qb][vt][4 * g + 1] * inv);
                        w.y = pack_bf16x2(O[qb][vt][4 * g + 2] * inv, O[qb][vt][4 * g + 3] * inv);
                        *(uint2*)(orow + d0) = w;
                    }
                }
            if (h == 0 && p.lse) p.lse[((size_t)b * p.H + head) * p.N + q] = (m[qb] + log2f(ltot)) * 0.6931471805599453f;
        }
    }
}

// =============================================================================================
// forward, matrix and vector work interleaved INSIDE each wave (long sequences, d = 40)
// =============================================================================================
// In attn_fwd_kernel a wave's stream is phase after phase -- 12 MFMAs (Q K^T), ~210 vector instructions (softmax), 16 MFMAs
// (P V) -- each waiting for the one before, and the SIMD's second wave drifts through the same phases at the same time: a
// tile costs a SIMD about the SUM of its matrix time (28 x 32 cycles) and its vector issue time (~1100 cycles), 2200-2300
// cycles measured.  Here the wave is software-pipelined over the key tiles so that every step holds three
// INDEPENDENT pieces of work -- P V of tile t-1, the softmax of tile t, Q K^T of tile t+1 -- and the step's 28 MFMAs are
// issued one fragment at a time between slices of the softmax (two or three exponent pairs per fragment, order pinned by
// sched_barrier): the matrix pipe runs under the vector stream of the SAME wave, whatever its partner does.  The step is
// then bound by vector issue alone (160 instructions + 28 MFMA issue slots), not by the sum.
//   LDS: two buffers, each K | V.  Step t reads K(t+1) and V(t-1) from buffer (t+1)&1 and writes K(t+2) and V(t) into buffer
//   t&1 (their previous contents were last read in step t-1, one barrier ago).  One barrier per step.
//   The reference point's rare move (MAX_SLACK) is decided before the exponents, as in attn_fwd_kernel; the rescale of O is
//   applied at the END of the step, after P V of tile t-1 -- whose P still carries the old reference -- has been issued.
//   Requires the spare "ones" column of V (the softmax denominator comes out of the P V product): d = 40.
template <int KS, int VT, int QB, int NW>
__global__ __launch_bounds__(64 * NW, (QB == 1 && NW == 4) ? 2 : 1) void attn_fwd_il_kernel(AttnParams p) {
    constexpr int NT = 64 * NW;                         // NW waves share the K / V tiles: 32 QB NW queries per workgroup
    using G = TileGeom<KS>;
    constexpr int KSTRIDE = G::RSTRIDE;
    constexpr int VSTRIDE = VGeom<VT>::VSTRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 64 * KSTRIDE + 64 * VSTRIDE;
    char* sKV = smem;                                   // [2][ K [64][KSTRIDE] | V [64][VSTRIDE] ]
    float* sBiasAll = (float*)(smem + 2 * TILE);        // [2][68]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    int bx, bh, bz_;
    attn_wg_coords(p, 1, bx, bh, bz_);
    const int b = bh / p.H, head = bh - b * p.H;
    const int q0 = bx * (32 * NW * QB) + wave * (32 * QB) + c;      // query of block qb: q0 + 32 * qb
    const int d = p.d;
    const float cs = p.scale * 1.4426950408889634f;

    bf16x8 qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            int ch = 2 * s + h, q = q0 + 32 * qb;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < p.N && ch * 8 < d) v = *(const uint4*)(p.q + ((size_t)b * p.N + q) * p.ldq + head * d + ch * 8);
            qf[qb][s] = __builtin_bit_cast(bf16x8, v);
        }
    f32x16 O[QB][VT];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[qb][vt][r] = 0.f;
    float m[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) m[qb] = -INFINITY;

    const uint16_t* kb = p.k + (size_t)b * p.M * p.ldk + head * d;
    const uint16_t* vb = p.v + (size_t)b * p.M * p.ldv + head * d;
    const int Mb = attn_key_count(p, b);
    const int ntiles = (Mb + 63) / 64;
    // global -> register -> LDS staging TWO steps deep (two register sets): the loads issued in step t are written to LDS at
    // the end of step t+1.  One step (~1 us) does not cover an L2 round trip under load -- measured with the loads removed:
    // 197 -> 130 us -- and the wait for them sat at the end of every step
    TileRegs<G::NCH, NT> rKa, rVa, rKb, rVb;
    TileMap<G::NCH, NT> mapK, mapV;
    mapK.init(p.ldk, KSTRIDE, d, tid);
    mapV.init(p.ldv, VSTRIDE, d, tid, d >> 3);          // the ones column (the launcher checked that it exists)
    // the key mask byte of a tile's key `lane` is loaded by every wave, without a branch and BEFORE the step's tile loads (a
    // younger load could only be waited for by draining those: vmcnt counts in order); wave 0 turns it into the tile's bias row
    const unsigned char* km = p.kmask ? p.kmask + (size_t)b * p.M : (const unsigned char*)p.q;
    auto mask_byte = [&](int key0) -> unsigned { return km[min(key0 + lane, p.M - 1)]; };
    auto key_bias = [&](int key0, float* sBias, unsigned mbyte) {
        if (tid < 64) {
            int key = key0 + tid;
            float bias = 0.f;
            if (key >= Mb) bias = -INFINITY;
            else if (p.kmask && !mbyte) bias = -FLT_MAX;
            sBias[tid] = bias;
            unsigned long long any = __ballot(bias != 0.f);
            if (tid == 0) sBias[64] = any ? 1.f : 0.f;
        }
    };
    // one operand fragment per item of a step: items 0 .. 4 VT - 1 are P V of the previous tile ((tt, sh) = (g >> 1, g & 1),
    // g = item / VT, d tile = item % VT: a transposed V fragment), the rest Q K^T of the next tile ((tt, s): a K fragment)
    constexpr int NPV = 4 * VT, NI = 4 * VT + 2 * KS;
    auto frag = [&](const char* sK, const char* sV, int i) -> bf16x8 {
        if (i < NPV) {
            const int g = i / VT, vt = i - g * VT;
            return lds_tr_frag(sV, VSTRIDE, 32 * (g >> 1) + 16 * (g & 1), 32 * vt, lane);
        }
        const int j = i - NPV, tt = j / KS, s = j - tt * KS;
        return *(const bf16x8*)(sK + (32 * tt + c) * KSTRIDE + (2 * s + h) * 16);
    };

    f32x16 SA[QB][2], SB[QB][2];
    uint4 PA[QB][2][2], PB[QB][2][2];                   // P fragments (bf16 pairs) of the previous / the current tile
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) PA[qb][tt][sh] = make_uint4(0, 0, 0, 0);

    // ---- prologue: K(0), K(1) in LDS, V(-1) = 0 (step 0's P V adds nothing), S(0) in registers
    tile_load_raw(rKa, mapK, kb, min(64, Mb));
    tile_store_fix(rKa, mapK, sKV, min(64, Mb));
    if (ntiles > 1) {
        tile_load_raw(rKa, mapK, kb + (size_t)64 * p.ldk, min(64, Mb - 64));
        tile_store_fix(rKa, mapK, sKV + TILE, min(64, Mb - 64));
    }
    // what "step -1" would have loaded: K(2) and V(0), stored at the end of step 0
    if (ntiles > 2) tile_load_raw(rKb, mapK, kb + (size_t)128 * p.ldk, min(64, Mb - 128));
    tile_load_raw(rVb, mapV, vb, min(64, Mb));
    for (int idx = tid; idx < 64 * VSTRIDE / 16; idx += NT) *(uint4*)(sKV + TILE + 64 * KSTRIDE + idx * 16) = make_uint4(0, 0, 0, 0);
    key_bias(0, sBiasAll, mask_byte(0));
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 kf = *(const bf16x8*)(sKV + (32 * tt + c) * KSTRIDE + (2 * s + h) * 16);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (s == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    SA[qb][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][s], z, 0, 0, 0);
                } else {
                    SA[qb][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][s], SA[qb][tt], 0, 0, 0);
                }
            }
        }
    __syncthreads();                                    // K(0) has been read by everyone before step 0 overwrites it with K(2)

    auto step = [&](int t, f32x16 (&Sc)[QB][2], f32x16 (&Sn)[QB][2], uint4 (&Pp)[QB][2][2], uint4 (&Pn)[QB][2][2],
                    TileRegs<G::NCH, NT>& rKl, TileRegs<G::NCH, NT>& rVl, const TileRegs<G::NCH, NT>& rKs,
                    const TileRegs<G::NCH, NT>& rVs) {
        const char* sK = sKV + ((t + 1) & 1) * TILE;   // K(t+1) | V(t-1)
        const char* sV = sK + 64 * KSTRIDE;
        const float* sBias = sBiasAll + (t & 1) * 68;
        const bool k2 = t + 2 < ntiles;
        const unsigned mbyte = mask_byte((t + 1) * 64);
        // loads of this step: K(t+3), V(t+1) -- stored at the end of the NEXT step
        // (past the last tile they re-read the last tile's rows and are never stored: no branch around a load)
        {
            const int tk = min(t + 3, ntiles - 1), tv = min(t + 1, ntiles - 1);
            tile_load_raw(rKl, mapK, kb + (size_t)tk * 64 * p.ldk, min(64, Mb - tk * 64));
            tile_load_raw(rVl, mapV, vb + (size_t)tv * 64 * p.ldv, min(64, Mb - tv * 64));
        }
        // ---- the tile's row maxima and the reference point (as attn_fwd_kernel)
        const bool biased = (p.kmask != nullptr || (t + 1 == ntiles && (Mb & 63) != 0)) && sBias[64] != 0.f;
        if (biased) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float4 bv = *(const float4*)(sBias + 32 * tt + 8 * g + 4 * h);
                        Sc[qb][tt][4 * g] += bv.x; Sc[qb][tt][4 * g + 1] += bv.y;
                        Sc[qb][tt][4 * g + 2] += bv.z; Sc[qb][tt][4 * g + 3] += bv.w;
                    }
        }
        float alpha[QB];
        bool anyg = false;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, Sc[qb][tt][r]);
            mx = xor32_max(mx);
            const float mcand = mx * cs;
            const bool grow = mcand > m[qb] + MAX_SLACK;
            const float mnew = grow ? mcand : m[qb];
            alpha[qb] = __builtin_amdgcn_exp2f(m[qb] - mnew);          // 1 for the rows that keep their reference
            m[qb] = mnew;
            anyg = anyg || __any(grow);
        }
        // ---- exponents of tile t between the MFMAs of P V (t-1) and Q K^T (t+1), one fragment at a time
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NP = QB * 16;                     // pairs of scores per lane
        // operand fragments two items ahead of their MFMAs (a ring of three), each read pinned in a region of its own: left
        // inside the item's region the compiler sinks it to just above its use and every MFMA waits a full LDS round trip
        bf16x8 fr[3];
        fr[0] = frag(sK, sV, 0);
        fr[1] = frag(sK, sV, 1);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + 2 < NI) fr[(i + 2) % 3] = frag(sK, sV, i + 2);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 fcur = fr[i % 3];
#pragma unroll
            for (int pi = i * NP / NI; pi < (i + 1) * NP / NI; ++pi) {
                const int qb = pi / 16, rest = pi - qb * 16, tt = rest >> 3, r = 2 * (rest & 7);
                const float e0 = __builtin_amdgcn_exp2f(fmaf(Sc[qb][tt][r], cs, -m[qb]));
                const float e1 = __builtin_amdgcn_exp2f(fmaf(Sc[qb][tt][r + 1], cs, -m[qb]));
                unsigned w = pack_bf16x2(e0, e1);
                // pinned HERE: left to itself the compiler sinks the whole exponent stream to the fragments' first use, the
                // next step's P V -- i.e. out of the MFMAs' shadow it is placed in
                asm volatile("" : "+v"(w));
                const int wi = (r & 7) >> 1;
                if (wi == 0) Pn[qb][tt][r >> 3].x = w;
                else if (wi == 1) Pn[qb][tt][r >> 3].y = w;
                else if (wi == 2) Pn[qb][tt][r >> 3].z = w;
                else Pn[qb][tt][r >> 3].w = w;
            }
            if (i < NPV) {
                const int g = i / VT, vt = i - g * VT;
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    O[qb][vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fcur, __builtin_bit_cast(bf16x8, Pp[qb][g >> 1][g & 1]), O[qb][vt], 0, 0, 0);
            } else {
                const int j = i - NPV, tt = j / KS, s = j - tt * KS;
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if (s == 0) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        Sn[qb][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fcur, qf[qb][s], z, 0, 0, 0);
                    } else {
                        Sn[qb][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fcur, qf[qb][s], Sn[qb][tt], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (anyg) {                                     // wave-uniform, rare: O (P V of t-1 included) moves to the new reference
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[qb][vt][r] *= alpha[qb];
        }
        char* wb = sKV + (t & 1) * TILE;               // K(t+2) | V(t): their buffers' last readers passed the previous barrier
        if (k2) tile_store_fix(rKs, mapK, wb, min(64, Mb - (t + 2) * 64));             // (loaded in the previous step)
        tile_store_fix(rVs, mapV, wb + 64 * KSTRIDE, min(64, Mb - t * 64));
        if (t + 1 < ntiles) key_bias((t + 1) * 64, sBiasAll + ((t + 1) & 1) * 68, mbyte);
        __syncthreads();
    };
    // (pairs of steps in a loop without a branch between them, the odd last step after it: with `if (t + 1 < ntiles)` inside
    // the loop the register sets of the staging pipeline were shuffled at the join -- a v_mov of a register still being loaded
    // = s_waitcnt vmcnt(0) at the end of every second step)
    int t = 0;
    for (; t + 1 < ntiles; t += 2) {
        step(t, SA, SB, PA, PB, rKa, rVa, rKb, rVb);
        step(t + 1, SB, SA, PB, PA, rKb, rVb, rKa, rVa);
    }
    if (t < ntiles) step(t, SA, SB, PA, PB, rKa, rVa, rKb, rVb);
    {   // P V of the last tile
        const char* sV = sKV + ((ntiles - 1) & 1) * TILE + 64 * KSTRIDE;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                bf16x8 vf = lds_tr_frag(sV, VSTRIDE, 32 * (g >> 1) + 16 * (g & 1), 32 * vt, lane);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if (ntiles & 1) O[qb][vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, PB[qb][g >> 1][g & 1]), O[qb][vt], 0, 0, 0);
                    else O[qb][vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, PA[qb][g >> 1][g & 1]), O[qb][vt], 0, 0, 0);
                }
            }
    }
    // ---- epilogue (as in attn_fwd_kernel, denominator from the ones column)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q0 + 32 * qb;
        const int vt0 = d >> 5, rin = d & 31, hh = (rin >> 2) & 1, reg = (rin & 3) + 4 * (rin >> 3);
        float lv = 0.f;
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (vt == vt0 && r == reg) lv = O[qb][vt][r];
        const float ltot = __shfl(lv, c + 32 * hh, 64);
        const float inv = 1.0f / ltot;
        if (q < p.N) {
            uint16_t* orow = p.o + ((size_t)b * p.N + q) * p.ldo + head * d;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int d0 = 32 * vt + 8 * g + 4 * h;
                    if (d0 < d) {
                        uint2 w;
                        w.x = pack_bf16x2(O[qb][vt][4 * g] * inv, O[qb][vt][4 * g + 1] * inv);
                        w.y = pack_bf16x2(O[qb][vt][4 * g + 2] * inv, O[qb][vt][4 * g + 3] * inv);
                        *(uint2*)(orow + d0) = w;
                    }
                }
            if (h == 0 && p.lse) p.lse[((size_t)b * p.H + head) * p.N + q] = (m[qb] + log2f(ltot)) * 0.6931471805599453f;
        }
    }
}

// =============================================================================================
// backward, part 1: dQ (query-stationary)
// =============================================================================================
template <int KS, int VT, bool PRE>
__global__ __launch_bounds__(256, (KS <= 6 ? 2 : 1)) void attn_bwd_dq_kernel(AttnParams p) {
    using G = TileGeom<KS>;
    constexpr int KSTRIDE = G::RSTRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                       // [64][KSTRIDE]   rows + transposed reads
    char* sV = sK + 64 * KSTRIDE;          // [64][KSTRIDE]   row reads
    float* sBias = (float*)(sV + 64 * KSTRIDE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    int bx, bh, bz_;
    attn_wg_coords(p, 2, bx, bh, bz_);
    const int b = bh / p.H, head = bh - b * p.H;
    const int q = bx * 128 + wave * 32 + c;
    const int d = p.d;
    const float cs = p.scale * 1.4426950408889634f;
    constexpr bool pre = PRE;             // (a template parameter: see attn_fwd_kernel)

    // -delta rides in as the dP accumulator's initial value (exact in f32), so dS = P * dP' needs no subtraction
    bf16x8 qf[KS], dof[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        int ch = 2 * s + h;
        uint4 v = make_uint4(0, 0, 0, 0), g = make_uint4(0, 0, 0, 0);
        if (q < p.N && ch * 8 < d) {
            v = *(const uint4*)(p.q + ((size_t)b * p.N + q) * p.ldq + head * d + ch * 8);
            g = *(const uint4*)(p.dout + ((size_t)b * p.N + q) * p.lddo + head * d + ch * 8);
        }
        qf[s] = __builtin_bit_cast(bf16x8, v);
        dof[s] = __builtin_bit_cast(bf16x8, g);
    }
    // delta[b][h][n] = sum_d dO * O of this query row, computed here (the lane pair l, l ^ 32 holds the even / odd 8-channel
    // chunks of dO already; O costs KS more 16-byte loads) and left in p.delta for the dK/dV kernel that follows on the
    // stream -- a separate pass over O and dO was one more launch per attention layer (64 per micro-batch)
    float lse2 = 0.f, dl = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int ch = 2 * s + h;
        if (q < p.N && ch * 8 < d) {
            const uint4 ov = *(const uint4*)(p.o + ((size_t)b * p.N + q) * p.ldo + head * d + ch * 8);
            float fo[8], fg[8];
            unpack_bf16x8(ov, fo);
            unpack_bf16x8(__builtin_bit_cast(uint4, dof[s]), fg);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += fo[e] * fg[e];
        }
    }
    dl += __shfl_xor(dl, 32, 64);
    if (q < p.N) {
        lse2 = p.lse[((size_t)b * p.H + head) * p.N + q] * 1.4426950408889634f;
        if (h == 0) p.delta[((size_t)b * p.H + head) * p.N + q] = dl;
    }
    f32x16 dQ[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQ[vt][r] = 0.f;

    const uint16_t* kb = p.k + (size_t)b * p.M * p.ldk + head * d;
    const uint16_t* vb = p.v + (size_t)b * p.M * p.ldv + head * d;
    const int Mb = attn_key_count(p, b);          // this sample's key count
    const int ntiles = (Mb + 63) / 64;
    TileRegs<G::NCH> rK, rV;
    TileMap<G::NCH> mapK, mapV;
    mapK.init(p.ldk, KSTRIDE, d, tid);
    mapV.init(p.ldv, KSTRIDE, d, tid);
    auto key_bias = [&](int key0) {
        if (tid < 64) {
            int key = key0 + tid;
            float bias = 0.f;
            if (key >= Mb) bias = -INFINITY;
            else if (p.kmask && !p.kmask[(size_t)b * p.M + key]) bias = -FLT_MAX;
            sBias[tid] = bias;
            unsigned long long any = __ballot(bias != 0.f);
            if (tid == 0) sBias[64] = any ? 1.f : 0.f;
        }
    };
    tile_load(rK, mapK, kb, min(64, Mb));
    tile_load(rV, mapV, vb, min(64, Mb));
    tile_store(rK, mapK, sK);
    tile_store(rV, mapV, sV);
    key_bias(0);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const bool more = kt + 1 < ntiles;
        if (more) {
            const int key1 = (kt + 1) * 64;
            tile_load(rK, mapK, kb + (size_t)key1 * p.ldk, min(64, Mb - key1));
            tile_load(rV, mapV, vb + (size_t)key1 * p.ldv, min(64, Mb - key1));
        }
        const bool biased = sBias[64] != 0.f;           // wave-uniform
        // three passes over the tile's two 32-key halves, as in the dK/dV kernel: all S / dP products, both halves' exponent chains,
        // all dQ products -- 12 and 8 back-to-back MFMAs, two independent vector chains (dQ kernel 200 -> 189 us at N 4096)
        f32x16 S[2], dP[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float s0 = pre ? -lse2 : 0.f;          // pre-scaled q: -lse rides in as the accumulator's initial value
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[t][r] = s0; dP[t][r] = -dl; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 kf = *(const bf16x8*)(sK + (32 * t + c) * KSTRIDE + (2 * s + h) * 16);
                S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], S[t], 0, 0, 0);
                bf16x8 vf = *(const bf16x8*)(sV + (32 * t + c) * KSTRIDE + (2 * s + h) * 16);
                dP[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[s], dP[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (biased) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 bv = *(const float4*)(sBias + 32 * t + 8 * g + 4 * h);
                    S[t][4 * g] += bv.x; S[t][4 * g + 1] += bv.y; S[t][4 * g + 2] += bv.z; S[t][4 * g + 3] += bv.w;
                }
            }
            if (pre) {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[t][r] = __builtin_amdgcn_exp2f(S[t][r]) * dP[t][r];                   // dS^T
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[t][r] = __builtin_amdgcn_exp2f(fmaf(S[t][r], cs, -lse2)) * dP[t][r];  // dS^T
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                bf16x8 dsf = acc_to_frag(S[t], sh);
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) {
                    bf16x8 ktr = lds_tr_frag(sK, KSTRIDE, 32 * t + 16 * sh, 32 * vt, lane);
                    dQ[vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktr, dsf, dQ[vt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) {
            tile_store(rK, mapK, sK);
            tile_store(rV, mapV, sV);
            key_bias((kt + 1) * 64);
            __syncthreads();
        }
    }
    if (q < p.N) {
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int d0 = 32 * vt + 8 * g + 4 * h;
                if (d0 < d) {
                    float o0 = dQ[vt][4 * g] * p.scale, o1 = dQ[vt][4 * g + 1] * p.scale;
                    float o2 = dQ[vt][4 * g + 2] * p.scale, o3 = dQ[vt][4 * g + 3] * p.scale;
                    if (p.tok_dt) {
                        const float* dtr = p.tok_dt + (((size_t)b * p.H + head) * p.N + q) * p.tok_G;
                        const float* kwp = p.tok_kw + ((size_t)(b * p.H + head) * p.tok_G) * d + d0;
                        for (int tg = 0; tg < p.tok_G; ++tg) {
                            const float t = dtr[tg] * p.scale;
                            const float4 kq = *(const float4*)(kwp + (size_t)tg * d);
                            o0 += t * kq.x; o1 += t * kq.y; o2 += t * kq.z; o3 += t * kq.w;
                        }
                    }
                    size_t off = ((size_t)b * p.N + q) * p.lddq + head * d + d0;
                    if (p.dq32) *(float4*)(p.dq32 + off) = make_float4(o0, o1, o2, o3);
                    if (p.dq16) {
                        uint2 w;
                        w.x = pack_bf16x2(o0, o1);
                        w.y = pack_bf16x2(o2, o3);
                        *(uint2*)(p.dq16 + off) = w;
                    }
                }
            }
    }
}

// =============================================================================================
// backward, part 2: dK, dV (key-stationary: a wave owns 32 keys, the workgroup 128 keys)
// =============================================================================================
// MASKED = false (no key mask; a key count at most): a key's column of S / P / dS feeds only that key's own dK / dV column, so
// the rows beyond the count are not biased inside the loop at all -- their columns (whatever they hold) are zeroed at the end.
// (The compiler turns the wave-uniform `if (biased) S += bias` into 32 adds + 32 selects per tile step on every path.)
template <int KS, int VT, bool PRE, bool MASKED>
__global__ __launch_bounds__(256, (KS <= 4 ? 2 : 1)) void attn_bwd_dkv_kernel(AttnParams p) {
    using G = TileGeom<KS>;
    constexpr int QSTRIDE = G::RSTRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two sets of [ Q [64][QSTRIDE] | dO [64][QSTRIDE] | lse [64] (log2 domain) | delta [64] ]: tile qt+1 is written (from the
    // registers its global loads landed in) while other waves may still be multiplying tile qt -- ONE barrier per tile step
    constexpr int SET = 2 * 64 * QSTRIDE + 128 * 4;
    char* sQ = smem;
    char* sDO = sQ + 64 * QSTRIDE;
    float* sLse = (float*)(sDO + 64 * QSTRIDE);
    float* sDl = sLse + 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    int bx, bh, bz;
    attn_wg_coords(p, 4, bx, bh, bz);
    const int b = bh / p.H, head = bh - b * p.H;
    // Which 128 keys: rotated by the (batch, head) row.  With a key count the workgroups of the LAST key blocks have nothing to
    // do; workgroups are placed round-robin over the CUs in launch order, and with 32 key blocks per row an unrotated map puts
    // all the idle ones on the same CUs (256 = 8 x 32: measured -- no gain at all from 28 % fewer keys).
    const int kblk = (int)(((unsigned)bx + (unsigned)bh) % gridDim.x);
    const int key = kblk * 128 + wave * 32 + c;
    const int d = p.d;
    const float cs = p.scale * 1.4426950408889634f;
    constexpr bool pre = PRE;             // (a template parameter: see attn_fwd_kernel)
    const int Mb = attn_key_count(p, b);          // this sample's key count (rows beyond it get zeros)

    // pre-scaled queries with a spare K chunk (dim_head 40): -lse and -delta of a query row ride in through that chunk
    // (bf16_split3): the Q / dO tiles carry their three terms in it, K / V fragments 1.0 -- instead of 16 LDS reads of 16 bytes
    // per lane and tile step that load them as the accumulators' initial values
    const bool padstat = pre && (d >> 3) < G::NCH && !(p.xcd & 32);
    const int pad_ch = padstat ? (d >> 3) : -1;
    // this wave's 32 keys as B operands
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        int ch = 2 * s + h;
        uint4 a = make_uint4(0, 0, 0, 0), g = make_uint4(0, 0, 0, 0);
        if (key < Mb && ch * 8 < d) {
            a = *(const uint4*)(p.k + ((size_t)b * p.M + key) * p.ldk + head * d + ch * 8);
            g = *(const uint4*)(p.v + ((size_t)b * p.M + key) * p.ldv + head * d + ch * 8);
        }
        if (ch == pad_ch) a = g = make_uint4(BF16_ONES3_X, BF16_ONES3_Y, 0, 0);
        kf[s] = __builtin_bit_cast(bf16x8, a);
        vf[s] = __builtin_bit_cast(bf16x8, g);
    }
    float bias = 0.f;
    if (key >= Mb) bias = -INFINITY;
    else if (p.kmask && !p.kmask[(size_t)b * p.M + key]) bias = -FLT_MAX;
    const bool biased = MASKED && __any(bias != 0.f);    // wave-uniform

    f32x16 dK[VT], dV[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dK[vt][r] = 0.f; dV[vt][r] = 0.f; }

    const uint16_t* qb = p.q + (size_t)b * p.N * p.ldq + head * d;
    const uint16_t* dob = p.dout + (size_t)b * p.N * p.lddo + head * d;
    const float* lseb = p.lse + ((size_t)b * p.H + head) * p.N;
    const float* dlb = p.delta + ((size_t)b * p.H + head) * p.N;
    const int ntiles = (p.N + 63) / 64;
    TileRegs<G::NCH> rQ, rDO;
    TileMap<G::NCH> mapQ, mapDO;
    mapQ.init(p.ldq, QSTRIDE, d, tid, -1, pad_ch);
    mapDO.init(p.lddo, QSTRIDE, d, tid, -1, pad_ch);
    float rl = 0.f, rd = 0.f;
    auto row_stats_load = [&](int q0) {        // stored negated: -lse2 is the fma addend, -delta the dP accumulator's initial value
        if (tid < 64) {
            bool ok = q0 + tid < p.N;
            rl = ok ? -lseb[q0 + tid] * 1.4426950408889634f : 0.f;
            rd = ok ? -dlb[q0 + tid] : 0.f;
        }
    };
    // this workgroup's slice of the query tiles (the whole range unless the grid was too small to fill the chip)
    const int split = bz;
    const int qt0 = (int)((long)split * ntiles / p.qsplit);
    // a workgroup whose 128 keys all lie beyond the sample's count has nothing to accumulate: it only writes its zeros
    const int qt1 = kblk * 128 >= Mb ? qt0 : (int)((long)(split + 1) * ntiles / p.qsplit);
    tile_load(rQ, mapQ, qb + (size_t)qt0 * 64 * p.ldq, min(64, p.N - qt0 * 64));
    tile_load(rDO, mapDO, dob + (size_t)qt0 * 64 * p.lddo, min(64, p.N - qt0 * 64));
    row_stats_load(qt0 * 64);
    auto row_stats_store = [&](int off) {         // off: byte offset of the buffer set
        if (tid < 64) {
            if (padstat) {
                const uint2 sl = bf16_split3(rl), sd = bf16_split3(rd);
                *(uint4*)(sQ + off + tid * QSTRIDE + pad_ch * 16) = make_uint4(sl.x, sl.y, 0, 0);
                *(uint4*)(sDO + off + tid * QSTRIDE + pad_ch * 16) = make_uint4(sd.x, sd.y, 0, 0);
            } else {
                *(float*)((char*)sLse + off + tid * 4) = rl;
                *(float*)((char*)sDl + off + tid * 4) = rd;
            }
        }
    };
    tile_store(rQ, mapQ, sQ);
    tile_store(rDO, mapDO, sDO);
    row_stats_store(0);
    __syncthreads();
    // the query-tile loop in two compiled forms (PADSTAT: the score tiles start from the inline constant 0 -- as a run-time flag in
    // one loop body the zero start is 64 register writes per step)
    auto qtiles = [&](auto padc) {
        constexpr bool PADSTAT = decltype(padc)::value;
        for (int qt = qt0; qt < qt1; ++qt) {
            const bool more = qt + 1 < qt1;
            const int cur = ((qt - qt0) & 1) * SET;        // this step's buffer set (byte offset)
            const char* sQ = smem + cur;
            const char* sDO = sQ + 64 * QSTRIDE;
            const float* sLse = (const float*)(sDO + 64 * QSTRIDE);
            const float* sDl = sLse + 64;
            if (more) {
                const int q1 = (qt + 1) * 64;
                tile_load(rQ, mapQ, qb + (size_t)q1 * p.ldq, min(64, p.N - q1));
                tile_load(rDO, mapDO, dob + (size_t)q1 * p.lddo, min(64, p.N - q1));
                row_stats_load(q1);
            }
            // Three passes over the tile's two 32-query halves -- all S / dP products, then both halves' exponent / dS chains, then
            // all dV / dK products -- instead of one half after the other: the matrix pipe gets 12 and 16 back-to-back MFMAs and the
            // vector unit two independent chains at a time (the two waves of a SIMD then overlap phases of different kinds more often)
            f32x16 S[2], dP[2], nl[2];
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) {
                // rows of S / dP are queries 32qi + 8g + 4h + e, the column (lane) is this wave's key
                if constexpr (PADSTAT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { S[qi][r] = 0.f; dP[qi][r] = 0.f; nl[qi][r] = 0.f; }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float4 lv = *(const float4*)(sLse + 32 * qi + 8 * g + 4 * h);
                        float4 dv = *(const float4*)(sDl + 32 * qi + 8 * g + 4 * h);
                        nl[qi][4 * g] = lv.x; nl[qi][4 * g + 1] = lv.y; nl[qi][4 * g + 2] = lv.z; nl[qi][4 * g + 3] = lv.w;
                        if (pre) { S[qi][4 * g] = lv.x; S[qi][4 * g + 1] = lv.y; S[qi][4 * g + 2] = lv.z; S[qi][4 * g + 3] = lv.w; }
                        else { S[qi][4 * g] = 0.f; S[qi][4 * g + 1] = 0.f; S[qi][4 * g + 2] = 0.f; S[qi][4 * g + 3] = 0.f; }
                        dP[qi][4 * g] = dv.x; dP[qi][4 * g + 1] = dv.y; dP[qi][4 * g + 2] = dv.z; dP[qi][4 * g + 3] = dv.w;
                    }
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    bf16x8 qrow = *(const bf16x8*)(sQ + (32 * qi + c) * QSTRIDE + (2 * s + h) * 16);
                    S[qi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qrow, kf[s], S[qi], 0, 0, 0);
                    bf16x8 drow = *(const bf16x8*)(sDO + (32 * qi + c) * QSTRIDE + (2 * s + h) * 16);
                    dP[qi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(drow, vf[s], dP[qi], 0, 0, 0);
                }
            }
            bf16x8 cdo[VT], cq[VT];
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                cdo[vt] = lds_tr_frag(sDO, QSTRIDE, 0, 32 * vt, lane);
                cq[vt] = lds_tr_frag(sQ, QSTRIDE, 0, 32 * vt, lane);
            }
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) {
                if constexpr (MASKED) {
                    if (biased) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) S[qi][r] += bias;
                    }
                }
                if (pre) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float pv = __builtin_amdgcn_exp2f(S[qi][r]);
                        S[qi][r] = pv;                 // P
                        dP[qi][r] = pv * dP[qi][r];    // dS
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float pv = __builtin_amdgcn_exp2f(fmaf(S[qi][r], cs, nl[qi][r]));
                        S[qi][r] = pv;                 // P
                        dP[qi][r] = pv * dP[qi][r];    // dS
                    }
                }
            }
            // dV / dK products, the transposed dO / Q fragments of 16-query slice i + 1 read while slice i multiplies (slice 0's
            // were read in front of the exponent phase above)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qi = i >> 1, sh = i & 1;
                bf16x8 ndo[VT], nq[VT];
                if (i < 3) {
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) {
                        ndo[vt] = lds_tr_frag(sDO, QSTRIDE, 16 * (i + 1), 32 * vt, lane);
                        nq[vt] = lds_tr_frag(sQ, QSTRIDE, 16 * (i + 1), 32 * vt, lane);
                    }
                }
                bf16x8 pf = acc_to_frag(S[qi], sh);
                bf16x8 dsf = acc_to_frag(dP[qi], sh);
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) {
                    dV[vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cdo[vt], pf, dV[vt], 0, 0, 0);
                    dK[vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cq[vt], dsf, dK[vt], 0, 0, 0);
                }
                if (i < 3) {
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) { cdo[vt] = ndo[vt]; cq[vt] = nq[vt]; }
                }
            }
            if (more) {                                    // the other set: its last readers passed the previous barrier
                char* nQ = smem + (SET - cur);
                tile_store(rQ, mapQ, nQ);
                tile_store(rDO, mapDO, nQ + 64 * QSTRIDE);
                row_stats_store(SET - cur);
            }
            __syncthreads();
        }
    };
    if constexpr (pre) {
        if (padstat) qtiles(std::true_type{});
        else qtiles(std::false_type{});
    } else {
        qtiles(std::false_type{});
    }
    if constexpr (!MASKED) {
        if (bias != 0.f) {              // a key beyond the sample's count: softmax weight exactly 0 (bias = -inf)
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { dK[vt][r] = 0.f; dV[vt][r] = 0.f; }
        }
    }
    if (key < p.M) {
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int d0 = 32 * vt + 8 * g + 4 * h;
                if (d0 < d) {
                    size_t offk = ((size_t)b * p.M + key) * p.lddk + head * d + d0;
                    size_t offv = ((size_t)b * p.M + key) * p.lddv + head * d + d0;
                    float k0 = dK[vt][4 * g] * p.scale, k1 = dK[vt][4 * g + 1] * p.scale;
                    float k2 = dK[vt][4 * g + 2] * p.scale, k3 = dK[vt][4 * g + 3] * p.scale;
                    if (p.qsplit > 1) {
                        const size_t C = (size_t)p.H * d, rows = (size_t)p.B * p.M;
                        float* pk = p.part + (((size_t)split * 2) * rows + (size_t)b * p.M + key) * C + head * d + d0;
                        *(float4*)pk = make_float4(k0, k1, k2, k3);
                        *(float4*)(pk + rows * C) =
                            make_float4(dV[vt][4 * g], dV[vt][4 * g + 1], dV[vt][4 * g + 2], dV[vt][4 * g + 3]);
                        continue;
                    }
                    if (p.tok_dt) {
                        const float* wr = p.tok_w + ((size_t)b * p.M + key) * p.tok_G;
                        const float* gqp = p.tok_gq + ((size_t)(b * p.H + head) * p.tok_G) * d + d0;
                        for (int tg = 0; tg < p.tok_G; ++tg) {
                            const float t = wr[tg] * p.scale;
                            const float4 gv = *(const float4*)(gqp + (size_t)tg * d);
                            k0 += t * gv.x; k1 += t * gv.y; k2 += t * gv.z; k3 += t * gv.w;
                        }
                    }
                    if (p.dk32) *(float4*)(p.dk32 + offk) = make_float4(k0, k1, k2, k3);
                    if (p.dk16) {
                        uint2 w;
                        w.x = pack_bf16x2(k0, k1);
                        w.y = pack_bf16x2(k2, k3);
                        *(uint2*)(p.dk16 + offk) = w;
                    }
                    if (p.dv32)
                        *(float4*)(p.dv32 + offv) =
                            make_float4(dV[vt][4 * g], dV[vt][4 * g + 1], dV[vt][4 * g + 2], dV[vt][4 * g + 3]);
                    if (p.dv16) {
                        uint2 w;
                        w.x = pack_bf16x2(dV[vt][4 * g], dV[vt][4 * g + 1]);
                        w.y = pack_bf16x2(dV[vt][4 * g + 2], dV[vt][4 * g + 3]);
                        *(uint2*)(p.dv16 + offv) = w;
                    }
                }
            }
    }
}

// fixed-order sum of the query-split partials -> dK / dV in the requested dtypes (deterministic, no atomics)
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(AttnParams p) {
    const int C4 = p.H * p.d / 4;
    const size_t rows = (size_t)p.B * p.M, C = (size_t)p.H * p.d;
    const size_t total = rows * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / C4;
        const int c = (int)(i - row * C4) * 4;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int sp = 0;
            for (; sp + 3 < p.qsplit; sp += 4) {          // four partials requested before the first is added (same order of sums)
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(p.part + (((size_t)(sp + u) * 2 + which) * rows + row) * C + c);
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; sp < p.qsplit; ++sp) {
                float4 v = *(const float4*)(p.part + (((size_t)sp * 2 + which) * rows + row) * C + c);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            if (which == 0 && p.tok_dt) {
                const int b = (int)(row / p.M), head = c / p.d, cd = c - head * p.d;
                const float* wr = p.tok_w + row * p.tok_G;
                const float* gqp = p.tok_gq + ((size_t)(b * p.H + head) * p.tok_G) * p.d + cd;
                for (int tg = 0; tg < p.tok_G; ++tg) {
                    const float t = wr[tg] * p.scale;
                    const float4 gv = *(const float4*)(gqp + (size_t)tg * p.d);
                    acc.x += t * gv.x; acc.y += t * gv.y; acc.z += t * gv.z; acc.w += t * gv.w;
                }
            }
            float* o32 = which ? p.dv32 : p.dk32;
            uint16_t* o16 = which ? p.dv16 : p.dk16;
            const long ld = which ? p.lddv : p.lddk;
            if (o32) *(float4*)(o32 + row * ld + c) = acc;
            if (o16) {
                uint2 w;
                w.x = pack_bf16x2(acc.x, acc.y);
                w.y = pack_bf16x2(acc.z, acc.w);
                *(uint2*)(o16 + row * ld + c) = w;
            }
        }
    }
}

// Kernel-selection switches, read from the environment ONCE (ADAP_ATTN_PP / ADAP_ATTN_FORCE_PP / ADAP_ATTN_PP_PRIO /
// ADAP_ATTN_QB1 / ADAP_ATTN_DKV_QSPLIT) instead of with getenv() on every launch; tests and the tuning tools change them
// through adap_attention_set_debug.
struct AttnDebug {
    int pp_mode;      // 0 = default choice, 1 = ping-pong where it applies, 2 = ping-pong always, 3 / 4 = the interleaved
                      // forward with 8 / 4 waves per workgroup wherever it applies, 5 = attn_fwd_kernel always
    int pp_prio;      // ping-pong kernel: phase at raised priority, 1 = matrix (default), 2 = vector, 0 = neither
    int qb1;          // 1 = one query block per wave even at the 64 x 64 level
    int dkv_qsplit;   // 0 = heuristic, else the dK/dV kernel's query-split factor
    int xcd;          // AttnParams::xcd bits (ADAP_ATTN_XCD=0: the plain blockIdx map everywhere, for A/B runs; +16 / +32: the
                      // forward's reference point / the dK/dV kernel's row statistics NOT through the spare K chunk)
};
static AttnDebug& attn_debug() {
    static AttnDebug d = [] {
        AttnDebug v;
        v.pp_mode = getenv("ADAP_ATTN_FORCE_PP") ? 2 : (getenv("ADAP_ATTN_PP") ? 1 : 0);
        const char* e = getenv("ADAP_ATTN_FWD_MODE");
        if (e) v.pp_mode = atoi(e);
        e = getenv("ADAP_ATTN_PP_PRIO");
        v.pp_prio = e ? atoi(e) : 1;
        v.qb1 = getenv("ADAP_ATTN_QB1") ? 1 : 0;
        e = getenv("ADAP_ATTN_DKV_QSPLIT");
        v.dkv_qsplit = e ? atoi(e) : 0;
        e = getenv("ADAP_ATTN_XCD");
        v.xcd = e ? atoi(e) : 7;
        return v;
    }();
    return d;
}
extern "C" int adap_attention_set_debug(int pp_mode, int pp_prio, int qb1, int dkv_qsplit) {
    ADAP_REQUIRE(pp_mode >= -1 && pp_mode <= 5 && pp_prio >= -1 && pp_prio <= 2 && qb1 >= -1 && qb1 <= 1 && dkv_qsplit >= -1 &&
                 dkv_qsplit <= 16, ADAP_ERR_UNSUPPORTED, "attention_set_debug: %d %d %d %d", pp_mode, pp_prio, qb1, dkv_qsplit);
    AttnDebug& d = attn_debug();                  // -1 leaves a switch as it is
    if (pp_mode >= 0) d.pp_mode = pp_mode;
    if (pp_prio >= 0) d.pp_prio = pp_prio;
    if (qb1 >= 0) d.qb1 = qb1;
    if (dkv_qsplit >= 0) d.dkv_qsplit = dkv_qsplit;
    return ADAP_OK;
}

// query-split factor of the dK/dV kernel: key-stationary workgroups number ceil(M/128)*B*H, which is only 32 for the
// cross-attention layers (M = 77) -- far too few for 256 CUs -- so the query loop is cut into slices
static int dkv_qsplit(int B, int H, int N, int M, int d) {
    const long blocks = (long)((M + 127) / 128) * B * H;
    const int ntiles = (N + 63) / 64;
    if (const int q = attn_debug().dkv_qsplit) {                   // tuning override (tools/attn_bwd_count_probe.py)
        if (q >= 1 && q <= 16 && q <= ntiles) return q;
    }
    if (blocks >= 256 || ntiles < 2) return 1;
    long want = (512 + blocks - 1) / blocks;
    if (want > 16) want = 16;
    if (want > ntiles) want = ntiles;
    // the partials are written and read once more: keep them under 16 MB, otherwise (self-attention at 16x16 with
    // 1280 channels: 10 MB per slice) the extra HBM pass costs more than the idle CUs did (measured 71 vs 48 us)
    const long per_slice = 2L * B * M * H * d * 4;
    while (want > 1 && want * per_slice > (16L << 20)) --want;
    return (int)want;
}

// =============================================================================================
// host dispatch
// =============================================================================================
static void* g_attn_stamps = nullptr;
extern "C" int adap_attention_set_stamp_buffer(void* buf) { g_attn_stamps = buf; return ADAP_OK; }
static int g_attn_fwd_variant = 0;       // 1 / 2: attn_fwd_kernel with QB = 1 / 2 query blocks per wave; 3: ping-pong kernel

template <int KS, int VT, int QB, bool PRE>
static int launch_fwd_qp(const AttnParams& p, hipStream_t s) {
    size_t lds = 2 * (64 * TileGeom<KS>::RSTRIDE + 64 * VGeom<VT>::VSTRIDE) + 2 * 68 * 4;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipFuncSetAttribute((const void*)attn_fwd_kernel<KS, VT, QB, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((p.N + 128 * QB - 1) / (128 * QB), p.B * p.H);
    hipLaunchKernelGGL((attn_fwd_kernel<KS, VT, QB, PRE>), grid, dim3(256), lds, s, p);
    g_attn_fwd_variant = QB;
    return adap_check_launch("attn_fwd");
}

template <int KS, int VT, int QB>
static int launch_fwd_q(const AttnParams& p, hipStream_t s) {
    return p.pre ? launch_fwd_qp<KS, VT, QB, true>(p, s) : launch_fwd_qp<KS, VT, QB, false>(p, s);
}

template <int KS, int VT>
static int launch_fwd_pp(const AttnParams& p, hipStream_t s) {
    size_t lds = 2 * 64 * TileGeom<KS>::RSTRIDE + 3 * 64 * VGeom<VT>::VSTRIDE + 3 * 68 * 4;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipFuncSetAttribute((const void*)attn_fwd_pp_kernel<KS, VT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((p.N + 511) / 512, p.B * p.H);
    AttnParams pp = p;
    pp.stamps = (unsigned long long*)g_attn_stamps;          // diagnostic stamps (null in normal runs)
    pp.prio_mode = attn_debug().pp_prio;
    hipLaunchKernelGGL((attn_fwd_pp_kernel<KS, VT>), grid, dim3(512), lds, s, pp);
    g_attn_fwd_variant = 3;
    return adap_check_launch("attn_fwd (ping-pong)");
}

template <int KS, int VT, int QB, int NW>
static int launch_fwd_il(const AttnParams& p, hipStream_t s) {
    size_t lds = 2 * (64 * TileGeom<KS>::RSTRIDE + 64 * VGeom<VT>::VSTRIDE) + 2 * 68 * 4;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipFuncSetAttribute((const void*)attn_fwd_il_kernel<KS, VT, QB, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((p.N + 32 * NW * QB - 1) / (32 * NW * QB), p.B * p.H);
    hipLaunchKernelGGL((attn_fwd_il_kernel<KS, VT, QB, NW>), grid, dim3(64 * NW), lds, s, p);
    g_attn_fwd_variant = 3 + QB + (NW == 8 ? 2 : 0);
    return adap_check_launch("attn_fwd (interleaved)");
}

template <int KS, int VT>
static int launch_fwd(const AttnParams& p, hipStream_t s) {
    // the interleaved kernel: needs V's spare ones column (d = 40) and a long key loop to pipeline over
    if constexpr (KS <= 4) {
        const int mode = attn_debug().pp_mode;
        const bool ones_col = 32 * VT > p.d && (p.d >> 3) < TileGeom<KS>::NCH;
        if ((mode == 3 || mode == 4) && !p.pre && ones_col && p.M >= 256) {
            if (mode == 3) return launch_fwd_il<KS, VT, 1, 8>(p, s);
            return launch_fwd_il<KS, VT, 1, 4>(p, s);
        }
    }
    // The ping-pong kernel (long sequences, short heads) is parity-green and opt-in: under sustained load it measures 155-157 us
    // on B4 N4096 d40 against 153.5 us for the kernel below, and the training step is 0.4 % faster without it.
    if constexpr (KS <= 4) {
        const int pp_mode = p.pre ? 0 : attn_debug().pp_mode;          // (the ping-pong kernel has no pre-scaled form)
        if (pp_mode == 2 || (pp_mode == 1 && p.M >= 512 && (long)((p.N + 511) / 512) * p.B * p.H >= 192))
            return launch_fwd_pp<KS, VT>(p, s);
    }
    // two query blocks per wave when that still leaves >= 2 workgroups per CU's worth of work (the 64x64 level)
    if (KS <= 4 && (long)((p.N + 255) / 256) * p.B * p.H >= 512 && !attn_debug().qb1) return launch_fwd_q<KS, VT, 2>(p, s);
    return launch_fwd_q<KS, VT, 1>(p, s);
}

template <int KS, int VT, bool PRE, bool MASKED>
static void launch_dkv(dim3 grid, size_t lds, hipStream_t s, const AttnParams& p) {
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<KS, VT, PRE, MASKED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<KS, VT, PRE, MASKED>), grid, dim3(256), lds, s, p);
}

template <int KS, int VT>
static int launch_bwd(const AttnParams& p_in, hipStream_t s) {
    // The XCD-aware workgroup map (attn_wg_coords) for the two backward kernels only where a (batch, head) row has few blocks:
    // at N = 4096 (32 blocks per row) the kernels are bound by vector / matrix issue, the Infinity Cache absorbs the re-fetches
    // of the plain map (FETCH 221 -> 66 MB per dK/dV launch with the map, time 511 -> 519 us), and a row's 32 workgroups walking
    // the same K / V (Q / dO) lines in step on one XCD cost more than the traffic saved; at 32 x 32 and below (<= 8 blocks per
    // row) the map takes 8-14 % off (profiles/r05_attn_xcd.md).  ADAP_ATTN_XCD bit 3 forces the map for every shape.
    AttnParams p = p_in;
    if (!(attn_debug().xcd & 8) && (p.N + 127) / 128 > 16) p.xcd &= ~6;
    size_t lds1 = 2 * 64 * TileGeom<KS>::RSTRIDE + 68 * 4;
    dim3 g1((p.N + 127) / 128, p.B * p.H);
    if (p.pre) hipLaunchKernelGGL((attn_bwd_dq_kernel<KS, VT, true>), g1, dim3(256), lds1, s, p);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<KS, VT, false>), g1, dim3(256), lds1, s, p);
    size_t lds2 = 2 * (2 * 64 * TileGeom<KS>::RSTRIDE + 128 * 4);          // two buffer sets
    dim3 g2((p.M + 127) / 128, p.B * p.H, p.qsplit);
    if (p.kmask) {
        if (p.pre) launch_dkv<KS, VT, true, true>(g2, lds2, s, p);
        else launch_dkv<KS, VT, false, true>(g2, lds2, s, p);
    } else {
        if (p.pre) launch_dkv<KS, VT, true, false>(g2, lds2, s, p);
        else launch_dkv<KS, VT, false, false>(g2, lds2, s, p);
    }
    if (p.qsplit > 1) {
        long tot = (long)p.B * p.M * (p.H * p.d / 4);
        long nb = (tot + 255) / 256;
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, s, p);
    }
    return adap_check_launch("attn_bwd");
}

#define ATTN_DISPATCH(FN, p, s)                                                  \
    do {                                                                         \
        int d_ = (p).d;                                                          \
        if (d_ <= 16) return FN<1, 1>(p, s);                                     \
        if (d_ <= 32) return FN<2, 1>(p, s);                                     \
        if (d_ <= 48) return FN<3, 2>(p, s);                                     \
        if (d_ <= 64) return FN<4, 2>(p, s);                                     \
        if (d_ <= 80) return FN<5, 3>(p, s);                                     \
        if (d_ <= 96) return FN<6, 3>(p, s);                                     \
        if (d_ <= 128) return FN<8, 4>(p, s);                                    \
        if (d_ <= 160) return FN<10, 5>(p, s);                                   \
        return adap_set_error(ADAP_ERR_UNSUPPORTED, "attention: dim_head %d > 160", d_); \
    } while (0)

static int attn_common_checks(const char* who, int B, int H, int N, int M, int d, long ldq, long ldk, long ldv) {
    ADAP_REQUIRE(B > 0 && H > 0 && N > 0 && M > 0, ADAP_ERR_SHAPE, "%s: empty problem", who);
    ADAP_REQUIRE(d % 8 == 0 && d >= 8 && d <= 160, ADAP_ERR_UNSUPPORTED, "%s: dim_head %d (need multiple of 8, <= 160)", who, d);
    ADAP_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0, ADAP_ERR_ALIGN, "%s: leading dims must be multiples of 8", who);
    ADAP_REQUIRE(ldq >= H * d && ldk >= H * d && ldv >= H * d, ADAP_ERR_SHAPE, "%s: leading dims < H*d", who);
    return ADAP_OK;
}

extern "C" int adap_attention_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                  const uint8_t* key_mask, const int* key_count, void* out, long ldo, float* lse,
                                  int B, int H, int N, int M, int d, float scale, void* stream) {
    ADAP_REQUIRE(q && k && v && out, ADAP_ERR_SHAPE, "attention_fwd: null pointer");
    int rc = attn_common_checks("attention_fwd", B, H, N, M, d, ldq, ldk, ldv);
    if (rc) return rc;
    ADAP_REQUIRE(ldo % 4 == 0 && ldo >= H * d, ADAP_ERR_ALIGN, "attention_fwd: ldo");
    AttnParams p = {};
    p.q = (const uint16_t*)q; p.ldq = ldq; p.k = (const uint16_t*)k; p.ldk = ldk; p.v = (const uint16_t*)v; p.ldv = ldv;
    p.kmask = key_mask; p.mcount = key_count; p.o = (uint16_t*)out; p.ldo = ldo; p.lse = lse;
    p.B = B; p.H = H; p.N = N; p.M = M; p.d = d; p.scale = scale; p.xcd = attn_debug().xcd;
    ADAP_REQUIRE(scale >= 0.f, ADAP_ERR_UNSUPPORTED, "attention_fwd: scale %g", (double)scale);
    if (scale == 0.f) {            // q carries d^-1/2 * log2(e) already (see the header): scores arrive in the exp2 domain
        p.pre = 1;
        p.scale = 0.6931471805599453f;          // 1 / log2(e): cs == 1 in the kernels
    }
    ATTN_DISPATCH(launch_fwd, p, (hipStream_t)stream);
}

extern "C" int adap_attention_fwd_last_variant(void) { return g_attn_fwd_variant; }

extern "C" long adap_attention_bwd_workspace_floats(int B, int H, int N, int M, int d) {
    long delta = ((long)B * H * N + 3) & ~3L;
    int qs = dkv_qsplit(B, H, N, M, d);
    return delta + (qs > 1 ? (long)qs * 2 * B * M * H * d : 0);
}

static int attention_bwd_impl(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                              const uint8_t* key_mask, const int* key_count, const void* out, long ldo, const void* dout,
                              long lddo, const float* lse, float* workspace,
                              float* dq32, void* dq16, long lddq, float* dk32, void* dk16, long lddk,
                              float* dv32, void* dv16, long lddv,
                              int B, int H, int N, int M, int d, float scale,
                              const float* tok_dt, const float* tok_w, const float* tok_prep, int tok_G, void* stream) {
    ADAP_REQUIRE(q && k && v && out && dout && lse && workspace, ADAP_ERR_SHAPE, "attention_bwd: null pointer");
    ADAP_REQUIRE(!tok_dt || (tok_w && tok_prep && tok_G >= 1 && tok_G <= TOK_MAXG && d % 4 == 0), ADAP_ERR_SHAPE,
                 "attention_bwd_tok: token-map arguments");
    float* delta_ws = workspace;
    ADAP_REQUIRE((dq32 || dq16) && (dk32 || dk16) && (dv32 || dv16), ADAP_ERR_SHAPE, "attention_bwd: missing output");
    int rc = attn_common_checks("attention_bwd", B, H, N, M, d, ldq, ldk, ldv);
    if (rc) return rc;
    ADAP_REQUIRE(ldo % 8 == 0 && lddo % 8 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0, ADAP_ERR_ALIGN,
                 "attention_bwd: leading dims");
    ADAP_REQUIRE(H * d / 8 <= 160 && H <= 64, ADAP_ERR_UNSUPPORTED, "attention_bwd: H*d too large");
    hipStream_t s = (hipStream_t)stream;
    AttnParams p = {};
    p.o = (uint16_t*)out; p.ldo = ldo;          // read only: the dQ kernel forms delta = rowsum(dO * O) from it
    p.q = (const uint16_t*)q; p.ldq = ldq; p.k = (const uint16_t*)k; p.ldk = ldk; p.v = (const uint16_t*)v; p.ldv = ldv;
    p.kmask = key_mask; p.mcount = key_count; p.lse = (float*)lse; p.dout = (const uint16_t*)dout; p.lddo = lddo;
    p.delta = delta_ws;
    p.dq32 = dq32; p.dq16 = (uint16_t*)dq16; p.lddq = lddq;
    p.dk32 = dk32; p.dk16 = (uint16_t*)dk16; p.lddk = lddk;
    p.dv32 = dv32; p.dv16 = (uint16_t*)dv16; p.lddv = lddv;
    p.B = B; p.H = H; p.N = N; p.M = M; p.d = d; p.scale = scale; p.xcd = attn_debug().xcd;
    ADAP_REQUIRE(scale >= 0.f && (scale > 0.f || !tok_dt), ADAP_ERR_UNSUPPORTED, "attention_bwd: scale %g", (double)scale);
    if (scale == 0.f) {            // pre-scaled q: dq is the gradient with respect to THAT q, i.e. ln 2 * dS K; dk = ln 2 * dS^T q
        p.pre = 1;
        p.scale = 0.6931471805599453f;
    }
    p.qsplit = dkv_qsplit(B, H, N, M, d);
    p.part = workspace + (((size_t)B * H * N + 3) & ~(size_t)3);
    if (tok_dt) {
        p.tok_dt = tok_dt; p.tok_w = tok_w; p.tok_G = tok_G;
        p.tok_kw = tok_prep;
        p.tok_gq = tok_prep + (size_t)B * H * tok_G * d;
    }
    ATTN_DISPATCH(launch_bwd, p, s);
}

extern "C" int adap_attention_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                  const uint8_t* key_mask, const int* key_count, const void* out, long ldo, const void* dout,
                                  long lddo, const float* lse, float* workspace,
                                  float* dq32, void* dq16, long lddq, float* dk32, void* dk16, long lddk,
                                  float* dv32, void* dv16, long lddv,
                                  int B, int H, int N, int M, int d, float scale, void* stream) {
    return attention_bwd_impl(q, ldq, k, ldk, v, ldv, key_mask, key_count, out, ldo, dout, lddo, lse, workspace, dq32, dq16, lddq,
                              dk32, dk16, lddk, dv32, dv16, lddv, B, H, N, M, d, scale, nullptr, nullptr, nullptr, 0, stream);
}

// adap_attention_bwd with the gradient of the layer's token maps (adap_attention_capture's side output) folded in:
// d_tokmap f32 [B][H][N][G], tok_w f32 [B][M][G], tok_prep = the workspace adap_attention_tokmap_prep filled for them.
extern "C" int adap_attention_bwd_tok(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                      const uint8_t* key_mask, const int* key_count, const void* out, long ldo, const void* dout,
                                      long lddo, const float* lse, float* workspace,
                                      float* dq32, void* dq16, long lddq, float* dk32, void* dk16, long lddk,
                                      float* dv32, void* dv16, long lddv,
                                      int B, int H, int N, int M, int d, float scale,
                                      const float* d_tokmap, const float* tok_w, const float* tok_prep, int G, void* stream) {
    return attention_bwd_impl(q, ldq, k, ldk, v, ldv, key_mask, key_count, out, ldo, dout, lddo, lse, workspace, dq32, dq16, lddq,
                              dk32, dk16, lddk, dv32, dv16, lddv, B, H, N, M, d, scale, d_tokmap, tok_w, tok_prep, G, stream);
}

// =============================================================================================
// Cross-attention side outputs for the distillation layers (attention.py:245-255): the pre-softmax
// scores (already scaled by dim_head^-0.5), the probabilities, and q * dim_head^-0.25.  Only the
// 12 cross-attention layers {7,8,12,16..24} with M <= 160 text tokens ask for them
// (openaimodel.py:947-952), so this is a small direct kernel: one wave per query row, lanes over
// keys, K^T of the (batch, head) in LDS.
// =============================================================================================
#define CAP_ROWS 16
__global__ __launch_bounds__(256) void attn_capture_kernel(const uint16_t* __restrict__ q, long ldq,
                                                           const uint16_t* __restrict__ k, long ldk,
                                                           float* __restrict__ score, float* __restrict__ prob,
                                                           float* __restrict__ qout, const float* __restrict__ tok_w,
                                                           float* __restrict__ tokmap, int G, int B, int H, int N, int M,
                                                           int d, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dp = d + 4;                  // padded row: float4 reads of consecutive keys land on different banks
    float* sK = (float*)smem;              // [M][dp]
    float* sQ = sK + M * dp;               // [4 waves][d]
    float* sW = sQ + 4 * d;                // [M][G] token weights (tokmap only)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int bh = blockIdx.y, b = bh / H, head = bh - b * H;
    if (tokmap)
        for (int idx = tid; idx < M * G; idx += 256) sW[idx] = tok_w[(size_t)b * M * G + idx];
    for (int idx = tid; idx < M * (d / 4); idx += 256) {
        int key = idx / (d / 4), c4 = idx - key * (d / 4);
        uint2 raw = *(const uint2*)(k + ((size_t)b * M + key) * ldk + head * d + 4 * c4);
        float4 v = make_float4(__builtin_bit_cast(float, raw.x << 16), __builtin_bit_cast(float, raw.x & 0xffff0000u),
                               __builtin_bit_cast(float, raw.y << 16), __builtin_bit_cast(float, raw.y & 0xffff0000u));
        *(float4*)(sK + key * dp + 4 * c4) = v;
    }
    __syncthreads();
    const float qs = sqrtf(scale);
    // 16 query rows per workgroup (4 per wave): the loop below is a chain of dependent LDS reads, so what it needs is
    // many waves per CU, not long ones -- with 64 rows the 16x16 level launched 128 four-wave workgroups, one per CU
    const int rows_per_block = CAP_ROWS;
    for (int rr = w; rr < rows_per_block; rr += 4) {
        const int n = blockIdx.x * rows_per_block + rr;
        if (n >= N) break;            // uniform per wave
        for (int dd = lane; dd < d; dd += 64) {
            float qv = bf16_to_f32(q[((size_t)b * N + n) * ldq + head * d + dd]);
            sQ[w * d + dd] = qv;
            if (qout) qout[(((size_t)b * H + head) * N + n) * d + dd] = qv * qs;
        }
        __builtin_amdgcn_s_waitcnt(0);    // this wave's LDS writes are visible to its own later reads
        __builtin_amdgcn_wave_barrier();
        float sv[3];
        float mx = -INFINITY;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            int key = lane + 64 * kk;
            float acc = -INFINITY;
            if (key < M) {
                const float4* kr = (const float4*)(sK + key * dp);
                const float4* qr = (const float4*)(sQ + w * d);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
                for (int c4 = 0; c4 < d / 4; ++c4) {
                    float4 kv = kr[c4], qv = qr[c4];
                    a0 += kv.x * qv.x; a1 += kv.y * qv.y; a2 += kv.z * qv.z; a3 += kv.w * qv.w;
                }
                acc = ((a0 + a1) + (a2 + a3)) * scale;
            }
            sv[kk] = acc;
            mx = fmaxf(mx, acc);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) sum += (lane + 64 * kk < M) ? __expf(sv[kk] - mx) : 0.f;
        sum = wave_sum(sum);
        const size_t base = (((size_t)b * H + head) * N + n) * M;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            int key = lane + 64 * kk;
            if (key < M) {
                if (score) score[base + key] = sv[kk];
                if (prob) prob[base + key] = __expf(sv[kk] - mx) / sum;
            }
        }
        if (tokmap) {       // sum over the listed tokens of this head's scores: sum_m score[n][m] * w[m][g]
            for (int g = 0; g < G; ++g) {
                float tsum = 0.f;
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    int key = lane + 64 * kk;
                    if (key < M) tsum += sv[kk] * sW[key * G + g];
                }
                tsum = wave_sum(tsum);
                if (lane == 0) tokmap[((((size_t)b * H + head) * N + n) * G) + g] = tsum;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Token maps alone (what the recon iteration's regularisers read of a layer, `dense=False` in ops.attention_capture):
// T[n][g] = sum_m score[n][m] w[m][g] is linear in the pre-softmax scores, so T[n][g] = <q[n], scale * kw[g]> with
// kw[g] = sum_m w[m][g] k[m] -- the factorisation the backward already uses (attn_tokmap_kw_kernel below).  No score row is
// formed: a workgroup builds the G x d vectors of its (batch, head) in LDS from the <= 192 keys (keys no group lists are
// skipped), then one thread per query row reads its d bf16 values and writes G floats.  HBM: q once (10 MB at 64 x 64, bs 4).
#define TOKF_ROWS 256
#define TOK_MAXP (4 * 160)                         // G <= 4 token groups x d <= 160
// kw[g][c] = sum_m w[m][g] k[m][c] of one (batch, head) into LDS, times `scale`.  With G d <= 256 outputs a thread per output
// walks all M keys while most of the workgroup idles (G d = 80 at d = 40): the keys are dealt over KS thread slices instead
// (item (s, p) sums the keys m = s, s + KS, ...) and the slices are summed in a fixed order -- measured 9.4 -> 5.0 us for the
// kw kernel and 15.5 -> 11.4 us for the forward at 64 x 64 (tools/microbench/tokmap_family.hip); KS = 1 is the plain walk.
// The keys' head slice [M][d] bf16 and the weights [M][G] are staged in LDS first (`stage`: M d bf16 + M G floats, dynamic): read
// straight from memory every thread walked the M keys with one 2-byte load per key -- a chain of ~10 cold-memory latencies
// (20-50 us inside the training step at d = 160, where G d = 320 outputs leave no slices to deal the keys over); staged, the
// whole slice arrives in one round of 8-byte loads.
__device__ __forceinline__ void tokmap_kw_phase(float* part, float* out_lds, char* stage, const float* __restrict__ w,
                                                const uint16_t* __restrict__ k, long ldk, int b, int head, int M, int d, int G,
                                                float scale) {
    uint16_t* sK = (uint16_t*)stage;                            // [M][d]
    float* sW = (float*)(stage + (((size_t)M * d * 2 + 15) & ~(size_t)15));   // [M][G]
    const int cpr = d >> 2;                                     // 8-byte chunks per key row (d % 4 == 0, rows 8-byte aligned: callers)
    for (int idx = threadIdx.x; idx < M * cpr; idx += 256) {
        const int m = idx / cpr, c4 = idx - m * cpr;
        *(uint2*)(sK + m * d + 4 * c4) = *(const uint2*)(k + ((size_t)b * M + m) * ldk + head * d + 4 * c4);
    }
    for (int idx = threadIdx.x; idx < M * G; idx += 256) sW[idx] = w[(size_t)b * M * G + idx];
    __syncthreads();
    const int P = G * d;
    int KS = 512 / P;
    KS = KS < 1 ? 1 : (KS > 8 ? 8 : KS);
    if (KS == 1) {
        for (int idx = threadIdx.x; idx < P; idx += 256) {
            const int g = idx / d, c = idx - g * d;
            float a = 0.f;
#pragma unroll 8
            for (int m = 0; m < M; ++m) a = fmaf(sW[m * G + g], bf16_to_f32(sK[m * d + c]), a);
            out_lds[idx] = a * scale;
        }
        __syncthreads();
        return;
    }
    for (int idx = threadIdx.x; idx < P * KS; idx += 256) {
        const int s = idx / P, p = idx - s * P, g = p / d, c = p - g * d;
        float a = 0.f;
        for (int m = s; m < M; m += KS) a = fmaf(sW[m * G + g], bf16_to_f32(sK[m * d + c]), a);
        part[idx] = a;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += 256) {
        float v = 0.f;
        for (int s = 0; s < KS; ++s) v += part[s * P + p];
        out_lds[p] = v * scale;
    }
    __syncthreads();
}
static size_t tokmap_kw_lds(int M, int d, int G) {          // (at least 4 KB: the forward kernel reuses the area for 256 x G row partials)
    const size_t n = (((size_t)M * d * 2 + 15) & ~(size_t)15) + (size_t)M * G * 4;
    return n < 4096 ? 4096 : n;
}
template <typename K>
static void tokmap_allow_lds(K kernel, size_t bytes) {          // beyond the 64 KB a kernel gets without asking (M = 192, d = 160)
    if (bytes > 48 * 1024) hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// The token maps of SEVERAL layers in one launch (adap_attention_tokmap_fwd_batched / _prep_batched): the 12 distillation layers'
// maps are only read by the losses after the UNet's forward, and their gradient prologue (kw, gq) only depends on what the forward
// saved and on the losses' gradients -- all known when the backward starts -- so neither has to sit on a block's dependency
// chain: one launch (grid.z = layer) behind the forward, three in front of the backward, instead of one / three per layer.
#define TOK_MAXL 16
struct TokLayer {
    const uint16_t* q; long ldq;
    const uint16_t* k; long ldk;
    const float* tok_w;         // [B][M][G]
    float* tokmap;              // forward: out [B][H][N][G]
    const float* dt;            // prep: d tokmap [B][H][N][G]
    float* ws;                  // prep: kw | gq | chunk partials (adap_attention_tokmap_prep_workspace_floats)
    int B, H, N, M, d, G;
    float scale;
};
struct TokBatch { TokLayer L[TOK_MAXL]; };

__device__ __forceinline__ void tokmap_fwd_body(const uint16_t* __restrict__ q, long ldq, const uint16_t* __restrict__ k, long ldk,
                                                const float* __restrict__ tok_w, float* __restrict__ tokmap, int G, int H, int N,
                                                int M, int d, float scale, int bx, int bh, float* sKW, float* sPart, char* tok_stage) {
    const int tid = threadIdx.x;
    const int b = bh / H, head = bh - b * H;
    tokmap_kw_phase(sPart, sKW, tok_stage, tok_w, k, ldk, b, head, M, d, G, scale);
    const int n = bx * TOKF_ROWS + tid;
    if (n >= N) return;
    const uint16_t* qr = q + ((size_t)b * N + n) * ldq + head * d;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // eight of the row's 8-byte chunks are requested before the first is used: one load per trip was a chain of d / 4 = 10 .. 40
    // cold-memory latencies per row inside the step (d % 4 == 0 and 8-byte aligned rows: checked by the caller)
    for (int c0 = 0; c0 < d; c0 += 32) {
        uint2 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) raw[u] = c0 + 4 * u < d ? *(const uint2*)(qr + c0 + 4 * u) : make_uint2(0u, 0u);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + 4 * u;
            if (c < d) {
                const float x0 = __builtin_bit_cast(float, raw[u].x << 16), x1 = __builtin_bit_cast(float, raw[u].x & 0xffff0000u);
                const float x2 = __builtin_bit_cast(float, raw[u].y << 16), x3 = __builtin_bit_cast(float, raw[u].y & 0xffff0000u);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (g < G) {
                        const float* kw = sKW + g * d + c;  // the same address in every lane: an LDS broadcast
                        acc[g] += (x0 * kw[0] + x1 * kw[1]) + (x2 * kw[2] + x3 * kw[3]);
                    }
            }
        }
    }
    float* out = tokmap + ((size_t)bh * N + n) * G;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (g < G) out[g] = acc[g];
}

__global__ __launch_bounds__(256) void attn_tokmap_fwd_kernel(const uint16_t* __restrict__ q, long ldq,
                                                              const uint16_t* __restrict__ k, long ldk,
                                                              const float* __restrict__ tok_w, float* __restrict__ tokmap,
                                                              int G, int H, int N, int M, int d, float scale) {
    __shared__ float sKW[TOK_MAXP];                 // [G <= 4][d <= 160]
    __shared__ float sPart[512];                    // the key slices' partial sums (G d <= 256: tokmap_kw_phase)
    extern __shared__ __attribute__((aligned(16))) char tok_stage[];
    tokmap_fwd_body(q, ldq, k, ldk, tok_w, tokmap, G, H, N, M, d, scale, blockIdx.x, blockIdx.y, sKW, sPart, tok_stage);
}

__global__ __launch_bounds__(256) void attn_tokmap_fwd_batched_kernel(TokBatch bt) {
    __shared__ float sKW[TOK_MAXP];
    __shared__ float sPart[512];
    extern __shared__ __attribute__((aligned(16))) char tok_stage[];
    const TokLayer& L = bt.L[blockIdx.z];
    if ((int)blockIdx.x * TOKF_ROWS >= L.N || (int)blockIdx.y >= L.B * L.H) return;          // (whole workgroups: before any barrier)
    tokmap_fwd_body(L.q, L.ldq, L.k, L.ldk, L.tok_w, L.tokmap, L.G, L.H, L.N, L.M, L.d, L.scale, blockIdx.x, blockIdx.y, sKW, sPart,
                    tok_stage);
}

extern "C" int adap_attention_capture(const void* q, long ldq, const void* k, long ldk, float* attnscore, float* attn,
                                      float* q_scaled, const float* tok_w, float* tokmap, int G, int B, int H, int N, int M,
                                      int d, float scale, void* stream) {
    ADAP_REQUIRE(q && k && (attnscore || attn || q_scaled || tokmap), ADAP_ERR_SHAPE, "attention_capture: null pointer");
    ADAP_REQUIRE(!tokmap || (tok_w && G >= 1 && G <= 4), ADAP_ERR_SHAPE, "attention_capture: token maps need weights, 1 <= G <= 4");
    ADAP_REQUIRE(M >= 1 && M <= 192, ADAP_ERR_UNSUPPORTED, "attention_capture: M=%d (cross-attention only, <= 192)", M);
    ADAP_REQUIRE(d >= 1 && d <= 160, ADAP_ERR_UNSUPPORTED, "attention_capture: d=%d", d);
    ADAP_REQUIRE(d % 4 == 0, ADAP_ERR_UNSUPPORTED, "attention_capture: d must be a multiple of 4");
    if (tokmap && !attnscore && !attn && !q_scaled && ldq % 4 == 0 && ((uintptr_t)q & 7) == 0 && ldk % 4 == 0 &&
        ((uintptr_t)k & 7) == 0 && (long)B * H <= 65535) {
        tokmap_allow_lds(attn_tokmap_fwd_kernel, tokmap_kw_lds(M, d, G));
        hipLaunchKernelGGL(attn_tokmap_fwd_kernel, dim3((N + TOKF_ROWS - 1) / TOKF_ROWS, B * H), dim3(256), tokmap_kw_lds(M, d, G),
                           (hipStream_t)stream, (const uint16_t*)q, ldq, (const uint16_t*)k, ldk, tok_w, tokmap, G, H, N, M, d,
                           scale);
        return adap_check_launch("attention_capture");
    }
    size_t lds = ((size_t)M * (d + 4) + 4 * d + (tokmap ? (size_t)M * G : 0)) * 4;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)attn_capture_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    dim3 grid((N + CAP_ROWS - 1) / CAP_ROWS, B * H);
    hipLaunchKernelGGL(attn_capture_kernel, grid, dim3(256), lds, (hipStream_t)stream, (const uint16_t*)q, ldq,
                       (const uint16_t*)k, ldk, attnscore, attn, q_scaled, tok_w, tokmap, G, B, H, N, M, d, scale);
    return adap_check_launch("attention_capture");
}

// =============================================================================================
// Gradient of the captured side outputs (the recon iteration's cross-layer consistency loss reads `attnscore` WITH
// gradient, ddpm.py:3246-3270, 4259-4387; stage 2 also `q`):
//   attnscore = scale * q k^T        ->  dq += scale * dS k,   dk += scale * dS^T q
//   q_scaled  = q * dim_head^-1/4    ->  dq += dim_head^-1/4 * dQs
// dq / dk are the bf16 gradients the flash backward has already written; the contributions are added in f32 and the
// sum is rounded once.  No atomics: every output element is owned by one thread (dq: one wave per query row, lanes
// over channels; dk: a workgroup per (batch, head, 16 keys), every thread walking all N queries).
// =============================================================================================
__global__ __launch_bounds__(256) void attn_capture_bwd_dq_kernel(const float* __restrict__ ds, const float* __restrict__ dqs,
                                                                  const uint16_t* __restrict__ k, long ldk,
                                                                  uint16_t* __restrict__ dq, long lddq, int B, int H, int N,
                                                                  int M, int d, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sK = (float*)smem;              // [M][d]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int bh = blockIdx.y, b = bh / H, head = bh - b * H;
    if (ds) {
        for (int idx = tid; idx < M * d; idx += 256) {
            int key = idx / d, c = idx - key * d;
            sK[idx] = bf16_to_f32(k[((size_t)b * M + key) * ldk + head * d + c]);
        }
    }
    __syncthreads();
    const float qs = sqrtf(scale);
    for (int rr = w; rr < CAP_ROWS; rr += 4) {
        const int n = blockIdx.x * CAP_ROWS + rr;
        if (n >= N) break;
        float acc[3] = {0.f, 0.f, 0.f};
        if (ds) {
            const float* row = ds + (((size_t)b * H + head) * N + n) * M;
            float dv[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) dv[kk] = (lane + 64 * kk < M) ? row[lane + 64 * kk] : 0.f;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int mbase = 64 * kk;
                if (mbase >= M) break;
                const int mend = min(64, M - mbase);
                for (int mm = 0; mm < mend; ++mm) {
                    const float g = __shfl(dv[kk], mm, 64);
                    const float* kr = sK + (mbase + mm) * d;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (lane + 64 * j < d) acc[j] += g * kr[lane + 64 * j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = lane + 64 * j;
            if (c < d) {
                float v = scale * acc[j];
                if (dqs) v += qs * dqs[(((size_t)b * H + head) * N + n) * d + c];
                uint16_t* o = dq + ((size_t)b * N + n) * lddq + head * d + c;
                *o = f32_to_bf16(bf16_to_f32(*o) + v);
            }
        }
    }
}

#define CAPB_KEYS 16
#define CAPB_ROWS 128
// dk, stage 1: partial[bh][chunk][m][c] = sum over the chunk's 128 query rows of dS[n][m] * q[n][c].  A workgroup owns
// (batch*head, row chunk, 16 keys); its dS slice sits in LDS (broadcast reads), q rows stream from global (64 lanes =
// 64 consecutive channels).  Cutting N into chunks is what fills the chip -- one workgroup per (batch*head, 16 keys)
// walking all 4096 rows with dependent loads took 4 ms per layer.  Stage 2 adds the chunks in fixed order.
__global__ __launch_bounds__(256) void attn_capture_bwd_dk_kernel(const float* __restrict__ ds, const uint16_t* __restrict__ q,
                                                                  long ldq, float* __restrict__ part, int B, int H, int N,
                                                                  int M, int d) {
    __shared__ float sD[CAPB_ROWS][CAPB_KEYS];
    const int tid = threadIdx.x, cl = tid & 63, kl = tid >> 6;          // 64 channel lanes x 4 key lanes
    const int bh = blockIdx.z, b = bh / H, head = bh - b * H;
    const int m0 = blockIdx.x * CAPB_KEYS, n0 = blockIdx.y * CAPB_ROWS;
    const int nchunks = gridDim.y;
    const float* dsb = ds + ((size_t)b * H + head) * N * M;
    for (int idx = tid; idx < CAPB_ROWS * CAPB_KEYS; idx += 256) {
        const int r = idx / CAPB_KEYS, mm = idx - r * CAPB_KEYS;
        const int n = n0 + r, m = m0 + mm;
        sD[r][mm] = (n < N && m < M) ? dsb[(size_t)n * M + m] : 0.f;
    }
    __syncthreads();
    float acc[4][3];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[kk][j] = 0.f;
    const uint16_t* qb = q + ((size_t)b * N + n0) * ldq + head * d;
    const int rows = min(CAPB_ROWS, N - n0);
#pragma unroll 4
    for (int r = 0; r < rows; ++r) {
        float qv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) qv[j] = (cl + 64 * j < d) ? bf16_to_f32(qb[(size_t)r * ldq + cl + 64 * j]) : 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float g = sD[r][kl + 4 * kk];
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[kk][j] += g * qv[j];
        }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int m = m0 + kl + 4 * kk;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = cl + 64 * j;
            if (c < d) part[(((size_t)bh * nchunks + blockIdx.y) * M + m) * d + c] = acc[kk][j];
        }
    }
}

// dk, stage 2: dk16[b][m][head*d + c] += scale * sum_chunk partial (fixed order; f32 sum rounded once)
__global__ __launch_bounds__(256) void attn_capture_bwd_dk_finish_kernel(const float* __restrict__ part, uint16_t* __restrict__ dk,
                                                                         long lddk, int B, int H, int M, int d, int nchunks,
                                                                         float scale) {
    const long total = (long)B * H * M * d;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % d);
        long r = i / d;
        const int m = (int)(r % M);
        const long bh = r / M;
        const int b = (int)(bh / H), head = (int)(bh - (long)b * H);
        float s = 0.f;
        for (int k = 0; k < nchunks; ++k) s += part[((bh * nchunks + k) * M + m) * d + c];
        uint16_t* o = dk + ((size_t)b * M + m) * lddk + head * d + c;
        *o = f32_to_bf16(bf16_to_f32(*o) + scale * s);
    }
}

// =============================================================================================
// Token maps: what the cross-layer consistency loss actually reads of attnscore is, per head, the sum over the
// subject (background) tokens, T[b][h][n][g] = sum_m attnscore[b][h][n][m] * w[b][m][g].  Its gradient never needs
// the dense [B][H][N][M] tensor:
//   dq[n] += scale * sum_g dT[n][g] * kw[g],   kw[g] = sum_m w[m][g] k[m]            (G vectors per (b, h))
//   dk[m] += scale * sum_g w[m][g]  * gq[g],   gq[g] = sum_n dT[n][g] q[n]           (G vectors per (b, h))
// gq is a reduction over the queries: 128-row chunks, two stages, fixed order.
// =============================================================================================
// kw[bh][g][c] = sum_m w[b][m][g] * k[b][m][head*d + c]  -- G*d numbers per (batch, head), one small workgroup each
__device__ __forceinline__ void tokmap_kw_body(const float* __restrict__ tok_w, const uint16_t* __restrict__ k, long ldk,
                                               float* __restrict__ kw, int H, int M, int d, int G, int bh, float* sKW, float* sPart,
                                               char* tok_stage) {
    const int b = bh / H, head = bh - b * H;
    tokmap_kw_phase(sPart, sKW, tok_stage, tok_w, k, ldk, b, head, M, d, G, 1.f);
    for (int p = threadIdx.x; p < G * d; p += 256) kw[(size_t)bh * G * d + p] = sKW[p];
}

__global__ __launch_bounds__(256) void attn_tokmap_kw_kernel(const float* __restrict__ tok_w, const uint16_t* __restrict__ k,
                                                             long ldk, float* __restrict__ kw, int H, int M, int d, int G) {
    __shared__ float sKW[TOK_MAXP];
    __shared__ float sPart[512];
    extern __shared__ __attribute__((aligned(16))) char tok_stage[];
    tokmap_kw_body(tok_w, k, ldk, kw, H, M, d, G, blockIdx.x, sKW, sPart, tok_stage);
}

__global__ __launch_bounds__(256) void attn_tokmap_kw_batched_kernel(TokBatch bt) {
    __shared__ float sKW[TOK_MAXP];
    __shared__ float sPart[512];
    extern __shared__ __attribute__((aligned(16))) char tok_stage[];
    const TokLayer& L = bt.L[blockIdx.y];
    if ((int)blockIdx.x >= L.B * L.H) return;
    tokmap_kw_body(L.tok_w, L.k, L.ldk, L.ws, L.H, L.M, L.d, L.G, blockIdx.x, sKW, sPart, tok_stage);
}

// dq[b][n][head*d + c] += scale * sum_g dT[b][head][n][g] * kw[bh][g][c]: element-wise, 8 channels per thread
__global__ __launch_bounds__(256) void attn_tokmap_bwd_dq_kernel(const float* __restrict__ dt, const float* __restrict__ kw,
                                                                 uint16_t* __restrict__ dq, long lddq, int B, int H, int N,
                                                                 int d, int G, float scale) {
    const int octs = d / 8;                                   // d % 8 == 0 (checked by the caller)
    const long total = (long)B * N * H * octs;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int o = (int)(i % octs);
        long r = i / octs;
        const int head = (int)(r % H);
        r /= H;
        const int n = (int)(r % N);
        const int b = (int)(r / N);
        const float* row = dt + ((((size_t)b * H + head) * N + n) * G);
        const float* kwp = kw + ((size_t)(b * H + head) * G) * d + 8 * o;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        for (int g = 0; g < G; ++g) {
            const float t = row[g];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t * kwp[g * d + e];
        }
        uint16_t* op = dq + ((size_t)b * N + n) * lddq + head * d + 8 * o;
        float cur[8];
        unpack_bf16x8(*(const uint4*)op, cur);
#pragma unroll
        for (int e = 0; e < 8; ++e) cur[e] += scale * v[e];
        *(uint4*)op = pack_bf16x8(cur);
    }
}

// stage 1 of gq: part[bh][chunk][g][c] = sum over the chunk's rows of dT[n][g] * q[n][c].  d / 8 lanes per query row, one 16-byte
// load each, RL = 256 / (d / 8) rows in flight per pass, then a fixed-order LDS reduction over the row lanes (one bf16 per lane
// with 40 of 64 lanes busy at d = 40 ran at 0.2 TB/s: 11.7 -> 6.2 us at 64 x 64, 14.7 -> 6.4 us at 16 x 16,
// tools/microbench/tokmap_family.hip).  Dynamic LDS: RL * G * d floats.
__device__ __forceinline__ void tokmap_gq_body(const float* __restrict__ dt, const uint16_t* __restrict__ q, long ldq,
                                               float* __restrict__ part, int H, int N, int d, int G, int bx, int nchunks, int bh,
                                               float* tok_red) {
    const int tid = threadIdx.x;
    const int octs = d >> 3, RL = 256 / octs;
    const int o = tid % octs, rl = tid / octs;
    const int b = bh / H, head = bh - b * H;
    const int n0 = bx * CAPB_ROWS;
    const int rows = min(CAPB_ROWS, N - n0);
    const int P = G * d;
    float acc[TOK_MAXG][8];
#pragma unroll
    for (int g = 0; g < TOK_MAXG; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    if (rl < RL) {
        const float* dtb = dt + (((size_t)b * H + head) * N + n0) * G;
        const uint16_t* qb = q + ((size_t)b * N + n0) * ldq + head * d + 8 * o;
        // four rows' loads are issued before the first is used: at d = 160 a lane walks 11 rows, and one load per trip was a
        // chain of 11 memory latencies (q is cold in the backward)
        for (int r0 = rl; r0 < rows; r0 += 4 * RL) {
            uint4 raw[4];
            float t[4][TOK_MAXG];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + u * RL;
                const bool ok = r < rows;
                raw[u] = ok ? *(const uint4*)(qb + (size_t)r * ldq) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int g = 0; g < TOK_MAXG; ++g) t[u][g] = (ok && g < G) ? dtb[(size_t)r * G + g] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float x[8];
                unpack_bf16x8(raw[u], x);
#pragma unroll
                for (int g = 0; g < TOK_MAXG; ++g)
                    if (g < G) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(t[u][g], x[e], acc[g][e]);
                    }
            }
        }
#pragma unroll
        for (int g = 0; g < TOK_MAXG; ++g)
            if (g < G) {
#pragma unroll
                for (int e = 0; e < 8; ++e) tok_red[(size_t)rl * P + g * d + 8 * o + e] = acc[g][e];
            }
    }
    __syncthreads();
    for (int p = tid; p < P; p += 256) {
        float v = 0.f;
        for (int i = 0; i < RL; ++i) v += tok_red[(size_t)i * P + p];
        part[((size_t)bh * nchunks + bx) * P + p] = v;
    }
}

__global__ __launch_bounds__(256) void attn_tokmap_bwd_gq_kernel(const float* __restrict__ dt, const uint16_t* __restrict__ q,
                                                                 long ldq, float* __restrict__ part, int B, int H, int N, int d,
                                                                 int G) {
    extern __shared__ float tok_red[];             // [RL][G * d]
    tokmap_gq_body(dt, q, ldq, part, H, N, d, G, blockIdx.x, gridDim.x, blockIdx.y, tok_red);
}

__global__ __launch_bounds__(256) void attn_tokmap_gq_batched_kernel(TokBatch bt) {
    extern __shared__ float tok_red[];
    const TokLayer& L = bt.L[blockIdx.z];
    const int nchunks = (L.N + CAPB_ROWS - 1) / CAPB_ROWS;
    if ((int)blockIdx.x >= nchunks || (int)blockIdx.y >= L.B * L.H) return;
    const long bhgd = (long)L.B * L.H * L.G * L.d;
    tokmap_gq_body(L.dt, L.q, L.ldq, L.ws + 2 * bhgd, L.H, L.N, L.d, L.G, blockIdx.x, nchunks, blockIdx.y, tok_red);
}
static size_t tokmap_gq_lds(int d, int G) { return (size_t)(256 / (d >> 3)) * G * d * sizeof(float); }

// stage 2: a workgroup per (batch*head, key): keys no group lists exit at once; the others sum the chunk partials of
// their channels in fixed order (every listed key repeats that small sum -- cheaper than a serial walk over the keys)
// and add scale * sum_g w[m][g] gq[g] into dk
__global__ __launch_bounds__(64) void attn_tokmap_bwd_dk_kernel(const float* __restrict__ part, const float* __restrict__ tok_w,
                                                                uint16_t* __restrict__ dk, long lddk, int B, int H, int M,
                                                                int d, int G, int nchunks, float scale) {
    const int m = blockIdx.x;
    const long bh = blockIdx.y;
    const int b = (int)(bh / H), head = (int)(bh - (long)b * H);
    float w[TOK_MAXG];
    bool any = false;
    for (int g = 0; g < TOK_MAXG; ++g) {
        w[g] = g < G ? tok_w[((size_t)b * M + m) * G + g] : 0.f;
        any = any || w[g] != 0.f;
    }
    if (!any) return;
    for (int c = threadIdx.x; c < d; c += 64) {
        float v = 0.f;
        for (int g = 0; g < G; ++g) {
            if (w[g] == 0.f) continue;
            float gq = 0.f;
            for (int k = 0; k < nchunks; ++k) gq += part[((bh * nchunks + k) * G + g) * d + c];
            v += w[g] * gq;
        }
        uint16_t* o = dk + ((size_t)b * M + m) * lddk + head * d + c;
        *o = f32_to_bf16(bf16_to_f32(*o) + scale * v);
    }
}

// gq[bh][g][c] = sum over the chunks of part[bh][chunk][g][c], fixed order
__global__ __launch_bounds__(256) void attn_tokmap_gq_reduce_kernel(const float* __restrict__ part, float* __restrict__ gq, int nchunks,
                                                                   int Gd, long total) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total) return;
    const long bh = i / Gd;
    const int e = (int)(i - bh * Gd);
    float a = 0.f;
#pragma unroll 8
    for (int k = 0; k < nchunks; ++k) a += part[(bh * nchunks + k) * Gd + e];
    gq[i] = a;
}

__global__ __launch_bounds__(256) void attn_tokmap_gq_reduce_batched_kernel(TokBatch bt) {
    const TokLayer& L = bt.L[blockIdx.y];
    const int Gd = L.G * L.d, nchunks = (L.N + CAPB_ROWS - 1) / CAPB_ROWS;
    const long total = (long)L.B * L.H * Gd;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total) return;
    const float* part = L.ws + 2 * total;
    const long bh = i / Gd;
    const int e = (int)(i - bh * Gd);
    float a = 0.f;
#pragma unroll 8
    for (int k = 0; k < nchunks; ++k) a += part[(bh * nchunks + k) * Gd + e];
    L.ws[total + i] = a;
}

// workspace floats of adap_attention_tokmap_prep: kw | gq | the gq chunk partials
extern "C" long adap_attention_tokmap_prep_workspace_floats(int B, int H, int N, int d, int G) {
    return (long)B * H * (((N + CAPB_ROWS - 1) / CAPB_ROWS) + 2) * G * d;
}

// The part of the token maps' backward that needs neither dq nor dk: kw = w^T K [B*H][G][d] (workspace + 0) and
// gq = dT^T Q [B*H][G][d] (workspace + B*H*G*d).  adap_attention_bwd_tok then adds scale * dT . kw into dq and
// scale * w . gq into dk inside its own epilogues -- no extra pass over dq / dk.
extern "C" int adap_attention_tokmap_prep(const float* d_tokmap, const float* tok_w, const void* q, long ldq, const void* k, long ldk,
                                          float* workspace, int B, int H, int N, int M, int d, int G, void* stream) {
    ADAP_REQUIRE(d_tokmap && tok_w && q && k && workspace, ADAP_ERR_SHAPE, "attention_tokmap_prep: null pointer");
    ADAP_REQUIRE(G >= 1 && G <= TOK_MAXG, ADAP_ERR_UNSUPPORTED, "attention_tokmap_prep: G=%d", G);
    ADAP_REQUIRE(d >= 1 && d <= 160 && d % 8 == 0 && M >= 1, ADAP_ERR_UNSUPPORTED, "attention_tokmap_prep: d=%d M=%d", d, M);
    ADAP_REQUIRE((long)B * H <= 65535, ADAP_ERR_SHAPE, "attention_tokmap_prep: B*H");
    hipStream_t s = (hipStream_t)stream;
    const int nchunks = (N + CAPB_ROWS - 1) / CAPB_ROWS;
    const long bhgd = (long)B * H * G * d;
    float* kw = workspace;
    float* gq = workspace + bhgd;
    float* part = workspace + 2 * bhgd;
    ADAP_REQUIRE(d % 4 == 0 && ldk % 4 == 0 && ((uintptr_t)k & 7) == 0, ADAP_ERR_ALIGN, "attention_tokmap: k rows must be 8-byte aligned");
    ADAP_REQUIRE(tokmap_kw_lds(M, d, G) <= 150 * 1024, ADAP_ERR_UNSUPPORTED, "attention_tokmap: M=%d keys x d=%d do not fit the LDS stage", M, d);
    tokmap_allow_lds(attn_tokmap_kw_kernel, tokmap_kw_lds(M, d, G));
    hipLaunchKernelGGL(attn_tokmap_kw_kernel, dim3(B * H), dim3(256), tokmap_kw_lds(M, d, G), s, tok_w, (const uint16_t*)k, ldk, kw, H, M, d, G);
    ADAP_REQUIRE(ldq % 8 == 0 && ((uintptr_t)q % 16) == 0, ADAP_ERR_ALIGN, "attention_tokmap_prep: q alignment");
    hipLaunchKernelGGL(attn_tokmap_bwd_gq_kernel, dim3(nchunks, B * H), dim3(256), tokmap_gq_lds(d, G), s, d_tokmap, (const uint16_t*)q,
                       ldq, part, B, H, N, d, G);
    hipLaunchKernelGGL(attn_tokmap_gq_reduce_kernel, dim3((unsigned)((bhgd + 255) / 256)), dim3(256), 0, s, part, gq, nchunks, G * d,
                       bhgd);
    return adap_check_launch("attention_tokmap_prep");
}

// n <= 16 layers described by flat host arrays: ptrs [n][6] = {q, k, tok_w, tokmap (fwd) or NULL, d_tokmap (prep) or NULL,
// workspace (prep) or NULL}; lds [n][2] = {ldq, ldk}; dims [n][6] = {B, H, N, M, d, G}; scales [n] (forward only).
static int tok_batch_fill(TokBatch& bt, int n, const void* const* ptrs, const long* lds, const int* dims, const float* scales,
                          bool fwd, const char* who) {
    ADAP_REQUIRE(n >= 1 && n <= TOK_MAXL && ptrs && lds && dims, ADAP_ERR_SHAPE, "%s: 1 <= layers <= %d", who, TOK_MAXL);
    for (int i = 0; i < n; ++i) {
        TokLayer& L = bt.L[i];
        L.q = (const uint16_t*)ptrs[6 * i]; L.k = (const uint16_t*)ptrs[6 * i + 1]; L.tok_w = (const float*)ptrs[6 * i + 2];
        L.tokmap = (float*)ptrs[6 * i + 3]; L.dt = (const float*)ptrs[6 * i + 4]; L.ws = (float*)ptrs[6 * i + 5];
        L.ldq = lds[2 * i]; L.ldk = lds[2 * i + 1];
        L.B = dims[6 * i]; L.H = dims[6 * i + 1]; L.N = dims[6 * i + 2]; L.M = dims[6 * i + 3]; L.d = dims[6 * i + 4]; L.G = dims[6 * i + 5];
        L.scale = scales ? scales[i] : 1.f;
        ADAP_REQUIRE(L.q && L.k && L.tok_w && (fwd ? (L.tokmap != nullptr) : (L.dt && L.ws)), ADAP_ERR_SHAPE, "%s: null pointer (layer %d)", who, i);
        ADAP_REQUIRE(L.B > 0 && L.H > 0 && L.N > 0 && (long)L.B * L.H <= 65535, ADAP_ERR_SHAPE, "%s: dims (layer %d)", who, i);
        ADAP_REQUIRE(L.G >= 1 && L.G <= TOK_MAXG && L.M >= 1 && L.M <= 192 && L.d >= 8 && L.d <= 160 && L.d % 8 == 0, ADAP_ERR_UNSUPPORTED,
                     "%s: G=%d M=%d d=%d (layer %d)", who, L.G, L.M, L.d, i);
        ADAP_REQUIRE(L.ldq % 8 == 0 && ((uintptr_t)L.q % 16) == 0 && L.ldk % 4 == 0 && ((uintptr_t)L.k & 7) == 0, ADAP_ERR_ALIGN,
                     "%s: q / k alignment (layer %d)", who, i);
        ADAP_REQUIRE(tokmap_kw_lds(L.M, L.d, L.G) <= 150 * 1024, ADAP_ERR_UNSUPPORTED, "%s: M=%d keys x d=%d do not fit the LDS stage", who, L.M, L.d);
    }
    return ADAP_OK;
}

// adap_attention_capture's token maps (no dense side outputs) for n layers in ONE launch: the same arithmetic per layer.
extern "C" int adap_attention_tokmap_fwd_batched(int n, const void* const* ptrs, const long* lds, const int* dims, const float* scales,
                                                 void* stream) {
    TokBatch bt = {};
    int rc = tok_batch_fill(bt, n, ptrs, lds, dims, scales, true, "attention_tokmap_fwd_batched");
    if (rc) return rc;
    unsigned gx = 0, gy = 0;
    size_t lds_b = 0;
    for (int i = 0; i < n; ++i) {
        const TokLayer& L = bt.L[i];
        gx = std::max(gx, (unsigned)((L.N + TOKF_ROWS - 1) / TOKF_ROWS));
        gy = std::max(gy, (unsigned)(L.B * L.H));
        lds_b = std::max(lds_b, tokmap_kw_lds(L.M, L.d, L.G));
    }
    tokmap_allow_lds(attn_tokmap_fwd_batched_kernel, lds_b);
    hipLaunchKernelGGL(attn_tokmap_fwd_batched_kernel, dim3(gx, gy, n), dim3(256), lds_b, (hipStream_t)stream, bt);
    return adap_check_launch("attention_tokmap_fwd_batched");
}

// adap_attention_tokmap_prep for n layers in THREE launches (kw, gq partials, gq): the same arithmetic per layer.
extern "C" int adap_attention_tokmap_prep_batched(int n, const void* const* ptrs, const long* lds, const int* dims, void* stream) {
    TokBatch bt = {};
    int rc = tok_batch_fill(bt, n, ptrs, lds, dims, nullptr, false, "attention_tokmap_prep_batched");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    unsigned bh_max = 0, ch_max = 0;
    size_t lds_kw = 0, lds_gq = 0;
    long tot_max = 0;
    for (int i = 0; i < n; ++i) {
        const TokLayer& L = bt.L[i];
        bh_max = std::max(bh_max, (unsigned)(L.B * L.H));
        ch_max = std::max(ch_max, (unsigned)((L.N + CAPB_ROWS - 1) / CAPB_ROWS));
        lds_kw = std::max(lds_kw, tokmap_kw_lds(L.M, L.d, L.G));
        lds_gq = std::max(lds_gq, tokmap_gq_lds(L.d, L.G));
        tot_max = std::max(tot_max, (long)L.B * L.H * L.G * L.d);
    }
    tokmap_allow_lds(attn_tokmap_kw_batched_kernel, lds_kw);
    hipLaunchKernelGGL(attn_tokmap_kw_batched_kernel, dim3(bh_max, n), dim3(256), lds_kw, s, bt);
    tokmap_allow_lds(attn_tokmap_gq_batched_kernel, lds_gq);
    hipLaunchKernelGGL(attn_tokmap_gq_batched_kernel, dim3(ch_max, bh_max, n), dim3(256), lds_gq, s, bt);
    hipLaunchKernelGGL(attn_tokmap_gq_reduce_batched_kernel, dim3((unsigned)((tot_max + 255) / 256), n), dim3(256), 0, s, bt);
    return adap_check_launch("attention_tokmap_prep_batched");
}

extern "C" long adap_attention_tokmap_bwd_workspace_floats(int B, int H, int N, int d, int G) {
    return (long)B * H * (((N + CAPB_ROWS - 1) / CAPB_ROWS) + 1) * G * d;          // gq partials + kw
}

extern "C" int adap_attention_tokmap_bwd(const float* d_tokmap, const float* tok_w, const void* q, long ldq, const void* k,
                                         long ldk, void* dq16, long lddq, void* dk16, long lddk, float* workspace, int B, int H,
                                         int N, int M, int d, int G, float scale, void* stream) {
    ADAP_REQUIRE(d_tokmap && tok_w && q && k && dq16 && dk16 && workspace, ADAP_ERR_SHAPE, "attention_tokmap_bwd: null pointer");
    ADAP_REQUIRE(G >= 1 && G <= TOK_MAXG, ADAP_ERR_UNSUPPORTED, "attention_tokmap_bwd: G=%d", G);
    ADAP_REQUIRE(d >= 1 && d <= 160 && M >= 1, ADAP_ERR_UNSUPPORTED, "attention_tokmap_bwd: d=%d M=%d", d, M);
    ADAP_REQUIRE((long)B * H <= 65535, ADAP_ERR_SHAPE, "attention_tokmap_bwd: B*H");
    hipStream_t s = (hipStream_t)stream;
    const int nchunks = (N + CAPB_ROWS - 1) / CAPB_ROWS;
    ADAP_REQUIRE(d % 8 == 0 && lddq % 8 == 0 && ((uintptr_t)dq16 % 16) == 0, ADAP_ERR_ALIGN, "attention_tokmap_bwd: dq alignment");
    float* kw = workspace + (size_t)B * H * nchunks * G * d;
    ADAP_REQUIRE(d % 4 == 0 && ldk % 4 == 0 && ((uintptr_t)k & 7) == 0, ADAP_ERR_ALIGN, "attention_tokmap: k rows must be 8-byte aligned");
    ADAP_REQUIRE(tokmap_kw_lds(M, d, G) <= 150 * 1024, ADAP_ERR_UNSUPPORTED, "attention_tokmap: M=%d keys x d=%d do not fit the LDS stage", M, d);
    tokmap_allow_lds(attn_tokmap_kw_kernel, tokmap_kw_lds(M, d, G));
    hipLaunchKernelGGL(attn_tokmap_kw_kernel, dim3(B * H), dim3(256), tokmap_kw_lds(M, d, G), s, tok_w, (const uint16_t*)k, ldk, kw, H, M, d, G);
    {
        const long tot = (long)B * N * H * (d / 8);
        long g = (tot + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(attn_tokmap_bwd_dq_kernel, dim3((unsigned)g), dim3(256), 0, s, d_tokmap, kw, (uint16_t*)dq16, lddq, B,
                           H, N, d, G, scale);
    }
    ADAP_REQUIRE(ldq % 8 == 0 && ((uintptr_t)q % 16) == 0, ADAP_ERR_ALIGN, "attention_tokmap_bwd: q alignment");
    hipLaunchKernelGGL(attn_tokmap_bwd_gq_kernel, dim3(nchunks, B * H), dim3(256), tokmap_gq_lds(d, G), s, d_tokmap, (const uint16_t*)q,
                       ldq, workspace, B, H, N, d, G);
    hipLaunchKernelGGL(attn_tokmap_bwd_dk_kernel, dim3(M, B * H), dim3(64), 0, s, workspace, tok_w, (uint16_t*)dk16, lddk, B, H, M,
                       d, G, nchunks, scale);
    return adap_check_launch("attention_tokmap_bwd");
}

extern "C" long adap_attention_capture_bwd_workspace_floats(int B, int H, int N, int M, int d) {
    return (long)B * H * ((N + CAPB_ROWS - 1) / CAPB_ROWS) * M * d;
}

extern "C" int adap_attention_capture_bwd(const float* d_attnscore, const float* d_q_scaled, const void* q, long ldq,
                                          const void* k, long ldk, void* dq16, long lddq, void* dk16, long lddk,
                                          float* workspace, int B, int H, int N, int M, int d, float scale, void* stream) {
    ADAP_REQUIRE((d_attnscore || d_q_scaled) && q && k && dq16, ADAP_ERR_SHAPE, "attention_capture_bwd: null pointer");
    ADAP_REQUIRE(!d_attnscore || (dk16 && workspace), ADAP_ERR_SHAPE, "attention_capture_bwd: d_attnscore needs dk and workspace");
    ADAP_REQUIRE(M >= 1 && M <= 192, ADAP_ERR_UNSUPPORTED, "attention_capture_bwd: M=%d (cross-attention only, <= 192)", M);
    ADAP_REQUIRE(d >= 1 && d <= 160, ADAP_ERR_UNSUPPORTED, "attention_capture_bwd: d=%d", d);
    ADAP_REQUIRE((long)B * H <= 65535, ADAP_ERR_SHAPE, "attention_capture_bwd: B*H");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = d_attnscore ? (size_t)M * d * 4 : 0;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)attn_capture_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(attn_capture_bwd_dq_kernel, dim3((N + CAP_ROWS - 1) / CAP_ROWS, B * H), dim3(256), lds, s, d_attnscore,
                       d_q_scaled, (const uint16_t*)k, ldk, (uint16_t*)dq16, lddq, B, H, N, M, d, scale);
    if (d_attnscore) {
        const int nchunks = (N + CAPB_ROWS - 1) / CAPB_ROWS;
        hipLaunchKernelGGL(attn_capture_bwd_dk_kernel, dim3((M + CAPB_KEYS - 1) / CAPB_KEYS, nchunks, B * H), dim3(256), 0, s,
                           d_attnscore, (const uint16_t*)q, ldq, workspace, B, H, N, M, d);
        const long total = (long)B * H * M * d;
        long g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(attn_capture_bwd_dk_finish_kernel, dim3((unsigned)g), dim3(256), 0, s, workspace, (uint16_t*)dk16, lddk,
                           B, H, M, d, nchunks, scale);
    }
    return adap_check_launch("attention_capture_bwd");
}
