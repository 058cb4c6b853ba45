// Weight-absorbed cross-attention of the decoder step (gfx950) - round 4.
//
// The reference feeds the encoder output to EVERY decoder call (TextDecoderInput.encoder_output_embeds, Core/Models.swift:970-1034,
// Core/TextDecoder.swift:381-418); the per-layer cross K / V rows are an implementation detail of the CoreML graph.  Streaming those
// rows is the dominant traffic of the step (large-v3, 64 slots: 491.5 MB per layer launch, 15.7 GB per session), so this path never
// materialises them.  With K_h = enc W_k,h^T and V_h = enc W_v,h^T + b_v,h:
//
//     q_h . K_h[t]      = (W_k,h^T q_h) . enc[t]                  = Q'_h . enc[t]           (k_proj has no bias)
//     sum_t p_t V_h[t]  = W_v,h (sum_t p_t enc[t]) + b_v,h        = W_v,h O'_h + b_v,h      (sum_t p_t = 1)
//
// so a layer reads the slot's encoder output [1500][d] f16 ONCE (half the bytes, the same tensor for all layers: 246 MB per session
// at 64 slots - Infinity-Cache sized) and the heads become the 32-wide side of MFMA tiles.  Three launches replace dec_cross_attn:
//
//   xabs_qk     Q'[slot][head][c] = sum_j W_k[h 64 + j][c] q[slot][h 64 + j]     (batch tile = N side, W_k^T tiles pre-tiled at load)
//   xabs_attn   one workgroup per (slot, key split): the split's encoder rows stream through an LDS ring by LDS-DMA (16-key tiles as
//               two 8-key half tiles of 20 KB at d = 1280, ring of 7 halves); per tile S^T = enc Q'^T on v_mfma_f32_16x16x32_f16 (the 8 waves split the channels,
//               partial tiles meet in LDS), online softmax with a deferred running maximum, P^T (f16) back through LDS,
//               O'^T += enc^T P^T on v_mfma_f32_32x32x16_f16 with the enc^T operand read by ds_read_b64_tr_b16; the split's
//               unnormalised O' [head][c] and (m, l) go to a partial buffer
//   xabs_vup    combines the splits while loading them as B fragments, att[slot][h 64 + j] = W_v,h O'_h / l + b_v -> the att planes
//               the cross out projection (dec32_proj RESID) already reads
//
// Everything is bit-deterministic and batch-invariant: a (slot, split) workgroup never looks at another slot, the split count depends
// on nothing but the build, partial sums are combined in index order.
#include "dec_shared.h"

namespace wh {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));

// 16-byte chunk c of tile row `key` lives in slot c ^ xswz(key) (low 4 bits): conflict-free for the S-phase ds_read_b128 (lane = key |
// k group << 4) AND for the transpose reads of the P V phase (4 consecutive keys x 4 consecutive chunks per 32-lane group)
__host__ __device__ __forceinline__ int xswz(int key) { return ((key & 3) << 2) | ((0x78 >> (2 * ((key >> 2) & 3))) & 3); }

// ---------------------------------------------------------------------------------------------- model-load helper
// W_k [d rows (h 64 + j)][d cols c] -> wkT[h][rt = c / 32][kt = j / 16][lane][8]: lane l = (c & 31) | (k half << 5) holds
// W_k[h 64 + kt 16 + 8 (l >> 5) + 0..7][rt 32 + (l & 31)] - the A fragment of Q'_h tile rows rt 32.. against q_h
__global__ void xabs_tile_wk_kernel(const f16* __restrict__ Wk, int d, int H, f16* __restrict__ out, size_t n_units) {
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_units) return;
    const int lane = (int)(o & 63);
    size_t t = o >> 6;
    const int kt = (int)(t & 3); t >>= 2;
    const int RT = d >> 5;
    const int rt = (int)(t % RT), h = (int)(t / RT);
    const int c = rt * 32 + (lane & 31), j0 = h * 64 + kt * 16 + 8 * (lane >> 5);
    f16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = Wk[(size_t)(j0 + i) * d + c];
    *reinterpret_cast<f16x8*>(out + o * 8) = v;
}
void xabs_tile_wk(const f16* Wk, int d, int H, f16* out, hipStream_t st) {
    const size_t n_units = (size_t)H * (d / 32) * 4 * 64;
    xabs_tile_wk_kernel<<<(unsigned)((n_units + 255) / 256), 256, 0, st>>>(Wk, d, H, out, n_units);
}

// ---------------------------------------------------------------------------------------------- xabs_qk
// grid (d / 256, H, n_bt), 4 waves, wave w: row tiles (blockIdx.x 4 + w) 2 + {0, 1}.  Output rows of d channels, f16, z ~ hi + lo / 2048:
//     qf_hi[slot][row 0..15]  = hi of heads 0..15        qf_lo[slot][row 0..15] = lo of heads 0..15
//     qf_hi[slot][row 16..23] = hi of heads 16..23       qf_hi[slot][row 24..31] = lo of heads 16..23        (NHT = 2 only)
// i.e. the second head tile of a 20-head model is PACKED: its (at most 8) real heads carry hi and lo in ONE 16-row tile, so the S phase
// of xabs_attn spends 3 MFMAs per k-step (tile 0 hi, tile 0 lo, tile 1 hi | lo) instead of 4 and keeps 60 instead of 80 fragment
// registers.  A wave writes whole 128-byte lines; xabs_attn gathers its S-phase B fragments (lane = row | k group << 4, 8 channels
// ks 32 + 8 kg + 0..7) with 64 contiguous bytes per row and k-step.
template <int NHT>
__global__ __launch_bounds__(256, 2) void xabs_qk_kernel(const XabsArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, bt = blockIdx.z;
    const int d = a.d, RT = d >> 5;
    const int j = lane & 31, hl = lane >> 5, gb = bt * 32 + j;
    // q_h as B fragments (K = 16 j-channels x N = 32 slots), hi | lo
    f16x8 qh[4], ql[4];
    {
        const float* qp = a.q + (size_t)gb * d + h * 64 + hl * 8;
        float4 v0[4], v1[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { v0[kt] = *reinterpret_cast<const float4*>(qp + kt * 16); v1[kt] = *reinterpret_cast<const float4*>(qp + kt * 16 + 4); }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const float z[8] = {v0[kt].x, v0[kt].y, v0[kt].z, v0[kt].w, v1[kt].x, v1[kt].y, v1[kt].z, v1[kt].w};
#pragma unroll
            for (int i = 0; i < 8; ++i) { f16 h_, l_; split_hilo(z[i], h_, l_); qh[kt][i] = h_; ql[kt][i] = l_; }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int rt = (blockIdx.x * 4 + wave) * 2 + s;
        const u32x4* wp = reinterpret_cast<const u32x4*>(a.wkT) + ((size_t)(h * RT + rt) * 4) * 64 + lane;
        u32x4 w[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) w[kt] = wp[kt * 64];
        f32x16 acc_h = {0}, acc_l = {0};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f16x8 wf = __builtin_bit_cast(f16x8, w[kt]);
            acc_h = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, qh[kt], acc_h, 0, 0, 0);
            acc_l = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, ql[kt], acc_l, 0, 0, 0);
        }
        // accumulator row (r & 3) + 8 (r >> 2) + 4 hl = channel c - rt 32; group g = r >> 2 is k group g, (r & 3) + 4 hl is the element
        // index: lanes l and l + 32 hold the two halves of one 16-byte unit.  v_permlane32_swap pairs them: afterwards the lower lane
        // owns the whole units of g = 0, 1, the upper lane those of g = 2, 3.
        unsigned ph[4][2], pl[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f16 zh[4], zl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) split_hilo(fmaf(acc_l[4 * g + i], 1.0f / 2048.0f, acc_h[4 * g + i]), zh[i], zl[i]);
            ph[g][0] = __builtin_bit_cast(unsigned, f16x2{zh[0], zh[1]});
            ph[g][1] = __builtin_bit_cast(unsigned, f16x2{zh[2], zh[3]});
            pl[g][0] = __builtin_bit_cast(unsigned, f16x2{zl[0], zl[1]});
            pl[g][1] = __builtin_bit_cast(unsigned, f16x2{zl[2], zl[3]});
        }
        u32x4 oh[2], ol[2];     // unit 0: g = 0 (lower lanes) / 2 (upper lanes); unit 1: g = 1 / 3
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // vdst = g = u (its upper half goes away), src0 = g = u + 2 (its lower half goes away)
                auto sh = __builtin_amdgcn_permlane32_swap(ph[u][i], ph[u + 2][i], false, false);
                auto sl = __builtin_amdgcn_permlane32_swap(pl[u][i], pl[u + 2][i], false, false);
                oh[u][i] = sh[0]; oh[u][2 + i] = sh[1];
                ol[u][i] = sl[0]; ol[u][2 + i] = sl[1];
            }
        }
        if (gb < a.batch) {
            // the lower lane holds channels rt 32 + 0..15 (units g = 0, 1), the upper lane rt 32 + 16..31: 32 contiguous bytes per lane and
            // plane, 128 contiguous bytes per slot over the wave's two row tiles
            const size_t o = ((size_t)gb * (NHT * 16) + h) * d + rt * 32 + 16 * hl;
            f16* lo_dst = h < 16 ? a.qf_lo + o : a.qf_hi + o + (size_t)8 * d;        // packed second tile: lo of head h lives 8 rows below its hi
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                *reinterpret_cast<u32x4*>(a.qf_hi + o + 8 * u) = oh[u];
                *reinterpret_cast<u32x4*>(lo_dst + 8 * u) = ol[u];
            }
        }
    }
    // cross-attention gate (dec_shared.h; opt-in): concurrent sessions take turns at the one kernel of a layer that saturates the HBM
    if (a.gate && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) xattn_gate_acquire(a.gate);
}

// ---------------------------------------------------------------------------------------------- S-phase fragments (shared)
// The Q' slice of one wave (k-steps wave CW + j) as B fragments of v_mfma_f32_16x16x32_f16: tile 0 hi, tile 0 lo, packed tile 1.
template <int KSW, int NHT>
struct XabsQFrag { f16x8 h0[KSW], l0[KSW], p1[NHT == 2 ? KSW : 1]; };
template <int KSW, int NHT>
__device__ __forceinline__ void xabs_load_qfrag(const XabsArgs& a, int b, int c0, int lane, XabsQFrag<KSW, NHT>& q) {
    const int row = lane & 15, H = a.n_head, D = a.d;
    const size_t qo = ((size_t)b * (NHT * 16) + row) * D + c0 + (lane >> 4) * 8;
    const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool real0 = row < H;                                            // padded heads: no request at all
    const bool real1 = NHT == 2 && 16 + (row & 7) < H;                     // packed tile: rows 0..7 hi, 8..15 lo of heads 16 + (row & 7)
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
        q.h0[j] = real0 ? *reinterpret_cast<const f16x8*>(a.qf_hi + qo + j * 32) : zero;
        q.l0[j] = real0 ? *reinterpret_cast<const f16x8*>(a.qf_lo + qo + j * 32) : zero;
        if constexpr (NHT == 2) q.p1[j] = real1 ? *reinterpret_cast<const f16x8*>(a.qf_hi + qo + (size_t)16 * D + j * 32) : zero;
    }
}
// S^T partial of one 16-key tile over this wave's channels: D[key = 4 (lane >> 4) + r][head = 16 ht + (lane & 15)] -> sreg[4 ht + r]
template <int KSW, int NHT>
__device__ __forceinline__ void xabs_s_tile(const f16x8 (&af)[KSW], const XabsQFrag<KSW, NHT>& q, int lane, float (&sreg)[NHT * 4]) {
    f32x4 sh = {0, 0, 0, 0}, sl = {0, 0, 0, 0}, sp = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
        sh = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j], q.h0[j], sh, 0, 0, 0);
        sl = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j], q.l0[j], sl, 0, 0, 0);
        if constexpr (NHT == 2) sp = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j], q.p1[j], sp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sreg[r] = fmaf(sl[r], 1.0f / 2048.0f, sh[r]);
    if constexpr (NHT == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lo = dpp_mov<kDppRor8>(sp[r]);             // column c + 8 of the packed tile: the lo part of head 16 + c
            sreg[4 + r] = (lane & 15) < 8 ? fmaf(lo, 1.0f / 2048.0f, sp[r]) : 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------- xabs_attn
constexpr int kXabsHalves = 7;               // LDS ring of half tiles (8 keys): tiles i, i + 1 resident, i + 2 and half of i + 3 in flight
constexpr int kXabsSpStride = 32 * 17;       // floats per wave partial: [32 heads][16 keys + 1]
constexpr float kXabsDefer = 8.0f;           // the running maximum moves only when it would grow by more than this (p <= e^8 fits f16)
__host__ __device__ constexpr int xabs_lds_bytes(int cw) { return kXabsHalves * cw * 4096 + 8 * kXabsSpStride * 4 + 1024 + 128; }

// (Round 5, measured and rejected, profiles/r05b_xabs_attn_l2_prefetch_ab_rejected.jsonl: an L2 PREFETCH ahead of the LDS ring - with every
// tile request a wave also touched one word of every 128-byte line of the half tile 2 or 3 tiles further on, a `buffer_load_dword ... lds`
// gather into a landing pad nobody reads, so that the ring's own requests become L2 hits and more than the ring's 60 KB per CU are in
// flight.  82.2 -> 92.3 us at 64 slots x 2 splits, 150.5 -> 172.5 us at 128 slots x 1 split, 2612 -> 2470 audio-s/s in flight: the CU's
// share of the stream is not bounded by the bytes it has in flight (latency), the extra requests compete for the same fetch path.)
template <int CW, int NHT, bool DBG, bool NTL>
__global__ __launch_bounds__(512, 2) void xabs_attn_kernel(const XabsArgs a) {
    constexpr int D = CW * 256, ROWB = D * 2, HALF = 8 * ROWB;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    float* spart = reinterpret_cast<float*>(smem + kXabsHalves * HALF);
    f16* pfrag = reinterpret_cast<f16*>(smem + kXabsHalves * HALF + 8 * kXabsSpStride * 4);
    float* alpha_l = reinterpret_cast<float*>(smem + kXabsHalves * HALF + 8 * kXabsSpStride * 4 + 1024);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: no waterfall loops around the buffer resource / M0
    // workgroup id -> (split, slot): id % 8 is the XCD; an XCD takes whole groups of 4 consecutive slots of one split, whose 32-byte
    // partial sectors share 128-byte lines (part[split][head][c / 8][slot][8]): the lines are assembled in one L2
    const int xr = blockIdx.x & 7, xq = blockIdx.x >> 3;
    const int S = a.n_split, H = a.n_head;
    // slots per workgroup (round 6): the grid covers the first `b1` slots; a workgroup then goes on to slots b + b1, b + 2 b1, ... - the share of
    // the CUs a launch takes (workgroups = b1 x splits) no longer grows with the batch, so a session's device batch can be larger than the half
    // of the chip its cross-attention is given.  Every slot is processed exactly as by a workgroup of its own: same bits.
    const int spw = max(a.spw, 1), b1 = (a.batch + spw - 1) / spw;
    const int n_grp = (b1 + 3) >> 2, grp = (xq >> 2) * 8 + xr;        // group = (split, 4 consecutive slots); group % 8 = its XCD
    const int sp = grp / n_grp, b_first = ((grp - sp * n_grp) << 2) + (xq & 3);
    if (a.gate && blockIdx.x == gridDim.x - 1 && tid == 0) xattn_gate_release(a.gate);     // the grid is draining from here on
    if (sp >= S || b_first >= b1) return;
    // WH_DBG=1: shader-clock stamps (tools/xabs_timeline.py): 9 entry, 10 slot state known, 11 loop entry, 12 loop exit, 13 partials stored
#define XPHASE(k) do { if constexpr (DBG) if (a.dbg && lane == 0 && (wave == 0 || wave == 5) && blockIdx.x < 64) \
        a.dbg[((blockIdx.x * 2 + (wave == 5)) * 2) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    constexpr int NT = (kCtx + 15) / 16;
    const int tile_lo = sp * NT / S, tile_hi = (sp + 1) * NT / S, n = tile_hi - tile_lo;
  for (int b = b_first; b < a.batch; b += b1) {            // (workgroup-uniform)
    if (b != b_first) __syncthreads();                    // every wave is done with the previous slot's ring, partial tiles and P^T before they are written again
    XPHASE(9);
    const int bc = a.cross_div > 1 ? b / a.cross_div : b;
    const unsigned char* enc = reinterpret_cast<const unsigned char*>(a.enc + (size_t)bc * kCtx * D);

    // ---- LDS-DMA: a 16-key tile is two half tiles of 8 keys; half k (= 2 tile + {0, 1}) lives in ring slot k % 7.  Waves 0-3 fetch the
    // first half of a tile, waves 4-7 the second (CW pieces of 1 KB each per wave and tile): the two halves of a tile can be requested
    // at different times, so that a freed tile (two slots) is refilled with the second half of tile i + 3 and the first half of
    // tile i + 4 - the fetch pipe of the CU (~11 B / clk: the bound of this kernel) always has a request queued behind the one it waits for.
    const int half_w = wave >> 2, wq = wave & 3;
    int p_off[CW];                       // byte offset of this lane's 16 bytes inside a tile's rows (loop-invariant)
#pragma unroll
    for (int p = 0; p < CW; ++p) {
        const int o = (wq * CW + p) * 1024 + lane * 16;
        const int row = o / ROWB, slot = (o - row * ROWB) >> 4;
        p_off[p] = (half_w * 8 + row) * ROWB + ((slot ^ xswz(half_w * 8 + row)) << 4);
    }
    int hi_mine = -1;                    // the last tile this wave has requested (wave-uniform)
    // buffer_load ... lds with a per-tile resource (base = the tile's first row, num_records = bytes to the end of the slot's encoder
    // output): the per-lane offset is loop-invariant, rows past position 1499 are out of range (nothing is fetched; they are masked
    // below), and a request costs two scalar instructions beside the load itself (a flat global_load_lds needs a 64-bit per-lane
    // address per piece: 600 - 800 cycles per tile and wave in the first version's timeline, profiles/r04g_*)
    auto issue = [&](int i) {
        unsigned char* dst = smem + ((2 * i + half_w) % kXabsHalves) * HALF + wq * (CW * 1024);
        const int t16 = (tile_lo + i) * 16;
#if defined(__HIP_DEVICE_COMPILE__)      // (the buffer-resource builtins do not exist in the host pass, which still has to emit this kernel's launch stub)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(enc) + (size_t)t16 * ROWB, 0, (kCtx - t16) * ROWB, 0x00020000);
#pragma unroll
        for (int p = 0; p < CW; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, p_off[p], 0, 0, NTL ? 2 : 0);
#else
        (void)dst; (void)t16; (void)p_off;
#endif
        hi_mine = i;
    };
    auto wait_tile = [&](int k) {        // this wave's pieces of tile k have landed; younger tiles stay in flight (requests return in order)
        switch (hi_mine - k) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CW) : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CW) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CW) : "memory"); break;
        }
    };
    // ---- prologue.  Request order = return order: the slot state first (so that the liveness test does not wait for the tile stream),
    // then the Q' fragments of this wave's channel slice (k-steps wave CW + j), then the first three and a half tiles.
    const SeqState* sq = a.seq + b;
    const int s_act = sq->active, s_done = sq->done, s_ti = sq->token_index;
    // softmax owner coordinates: lane = key | (head & 3) << 4, wave = head >> 2
    const int o_key = lane & 15, o_head = 4 * wave + (lane >> 4);
    const bool owner = o_head < 16 * NHT;
    int al_slot = -1;
    if (a.align && owner && o_head < H) al_slot = a.align_slot[a.layer * H + o_head];
    XabsQFrag<CW, NHT> qf;
    xabs_load_qfrag<CW, NHT>(a, b, wave * CW * 32, lane, qf);
    issue(0);
    if (n > 1) issue(1);
    if (n > 2) issue(2);
    if (n > 3 && half_w == 0) issue(3);
    if (tid < 32) alpha_l[tid] = 1.0f;
    pfrag[tid] = (f16)0.0f;
    if (!(s_act && !s_done)) {           // workgroup-uniform: a finished slot streams nothing more
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        continue;
    }
    const int pos = min(max(s_ti, 0), kMaxTok - 1);
    float* raw = nullptr;
    if (al_slot >= 0 && pos + 1 < kMaxTok) raw = a.align + (((size_t)b * kMaxTok + pos + 1) * a.n_align + al_slot) * kCtx;
    XPHASE(10);

    f32x16 acc[CW];
#pragma unroll
    for (int mt = 0; mt < CW; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    // S phase A operand: lane = key | k group << 4; P V phase transpose-read supplier: 16-lane group g16, supplier index s
    const int s_key = lane & 15, s_kg = lane >> 4, s_sw = xswz(s_key);
    const int g16 = lane >> 4, sl = lane & 15;
    const int t_key0 = (g16 >> 1) * 8 + (sl >> 2), t_key1 = t_key0 + 4;
    const int t_c = (g16 & 1) * 2 + ((sl & 3) >> 1), t_b = (sl & 1) * 8;
    const int t_off0 = (t_key0 & 7) * ROWB + t_b, t_off1 = (t_key1 & 7) * ROWB + t_b, t_sw0 = xswz(t_key0), t_sw1 = xswz(t_key1);
    // tile i, this lane's half as S-phase reader (key >> 3) and as transpose-read supplier (g16 >> 1)
    auto s_base = [&](int i) { return smem + ((2 * i + (s_key >> 3)) % kXabsHalves) * HALF + (s_key & 7) * ROWB; };
    auto t_base = [&](int i) { return smem + ((2 * i + (g16 >> 1)) % kXabsHalves) * HALF; };

    // S^T partial of one tile over this wave's channels, [16 keys] x [16 NHT heads]: D[key = 4 (lane >> 4) + r][head = lane & 15]
    auto s_phase = [&](const unsigned char* row, float (&sreg)[NHT * 4]) {
        f16x8 af[CW];
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            const int c = (wave * CW + j) * 4 + s_kg;
            af[j] = *reinterpret_cast<const f16x8*>(row + ((c ^ s_sw) << 4));
        }
        xabs_s_tile<CW, NHT>(af, qf, lane, sreg);
    };
    auto write_partials = [&](const float (&sreg)[NHT * 4]) {
        float* wpart = spart + wave * kXabsSpStride;
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
            for (int r = 0; r < 4; ++r) wpart[(ht * 16 + (lane & 15)) * 17 + 4 * (lane >> 4) + r] = sreg[ht * 4 + r];
    };
    // the owner lane of (key, head): sum of the 8 channel-slice partials, online softmax with a deferred running maximum
    auto softmax = [&](int i) {
        float s = 0.0f;
#pragma unroll
        for (int v = 0; v < 8; ++v) s += spart[v * kXabsSpStride + o_head * 17 + o_key];
        const int t = (tile_lo + i) * 16 + o_key;
        const bool valid = t < kCtx;
        if (raw && valid) raw[t] = s;                 // alignment heads: DecodingCache.alignmentWeights row tokenIndex + 1 (raw scores)
        s = valid ? s : -INFINITY;
        float mt_ = s;                                // maximum over the 16 keys = the 16 lanes of a DPP row
        mt_ = fmaxf(mt_, dpp_mov<kDppXor1>(mt_));
        mt_ = fmaxf(mt_, dpp_mov<kDppXor2>(mt_));
        mt_ = fmaxf(mt_, dpp_mov<kDppHalfMirror>(mt_));
        mt_ = fmaxf(mt_, dpp_mov<kDppMirror>(mt_));
        const float m_new = fmaxf(m_run, mt_);
        const float m_use = (m_new > m_run + kXabsDefer) ? m_new : m_run;     // (m_run = -inf: the first tile always takes its maximum)
        const float al = __expf(m_run - m_use);       // 1 when the maximum stays, 0 on the first tile
        float p = valid ? __expf(s - m_use) : 0.0f;
        if (o_head >= H) p = 0.0f;
        const f16 ph = (f16)p;
        float ps = (float)ph;                         // the denominator sums what the numerator multiplies
        ps += dpp_mov<kDppXor1>(ps);
        ps += dpp_mov<kDppXor2>(ps);
        ps += dpp_mov<kDppHalfMirror>(ps);
        ps += dpp_mov<kDppMirror>(ps);
        l_run = fmaf(l_run, al, ps);
        m_run = m_use;
        // P^T as the B fragment of the P V MFMA: lane' = head | (key >> 3) << 5, element key & 7
        pfrag[(o_head | ((o_key >> 3) << 5)) * 8 + (o_key & 7)] = ph;
        if (o_key == 0) alpha_l[o_head] = (o_head < H) ? al : 1.0f;
    };
    // O'^T[channel][head] += enc^T[channel][key] P^T[key][head]: A = enc^T tile (M = 32 channels, K = 16 keys), two transpose reads
    // O'^T[channel][head] += enc^T[channel][key] P^T[key][head]: A = enc^T tile (M = 32 channels, K = 16 keys), two transpose reads.
    // (Measured and rejected, profiles/r04z_*: issuing the transpose reads of tile i in the Y interval - they do not depend on P - costs
    // 57 -> 60 us: the barrier's lgkmcnt(0) waits for them anyway; softmax ahead of the S-phase MFMAs costs 57 -> 68 us: an in-order
    // wave stalls on the softmax's LDS round trip before it has queued its matrix work.)
    auto pv = [&](const unsigned char* tile) {
        const f16x8 pf = *reinterpret_cast<const f16x8*>(pfrag + lane * 8);
        const float al = alpha_l[lane & 31];
        if (__builtin_amdgcn_ballot_w64(al != 1.0f)) {        // wave-uniform, rare: some head's running maximum moved
#pragma unroll
            for (int mt = 0; mt < CW; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] *= al;
        }
#pragma unroll
        for (int mt = 0; mt < CW; ++mt) {
            const int c = (wave * CW + mt) * 4 + t_c;
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + t_off0 + ((c ^ t_sw0) << 4)));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + t_off1 + ((c ^ t_sw1) << 4)));
            const f16x4 f0 = __builtin_bit_cast(f16x4, a0), f1 = __builtin_bit_cast(f16x4, a1);
            const f16x8 af = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, pf, acc[mt], 0, 0, 0);
        }
    };

    // Software pipeline, two barriers per tile:  Y_i = softmax(i) (LDS / VALU latency chain of the owner lanes) beside S(i + 1) (MFMA);
    // Z_i = partials(i + 1) -> LDS, P V(i).  Tiles i (P V) and i + 1 (S) are resident, tile i + 2 and the first half of tile i + 3 are in
    // flight; the barrier that ends Z_i frees tile i's two half slots for the second half of tile i + 3 and the first half of tile i + 4.
    float sreg[NHT * 4];
    wait_tile(0);
    __syncthreads();
    s_phase(s_base(0), sreg);
    write_partials(sreg);
    if (n > 1) wait_tile(1);
    __syncthreads();                                      // partials(0) in LDS, tile 1 landed
    XPHASE(11);
    // WH_DBG=1: shader-clock stamps of tiles 10 and 11 for waves 0 and 5 of the first 64 workgroups (tools/xabs_timeline.py)
#define XSTAMP(k) do { if constexpr (DBG) if (a.dbg && (i == 10 || i == 11) && lane == 0 && (wave == 0 || wave == 5) && blockIdx.x < 64) \
        a.dbg[((blockIdx.x * 2 + (wave == 5)) * 2 + (i - 10)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    for (int i = 0; i < n; ++i) {
        XSTAMP(0);
        const bool work = !DBG || !(a.ablate & 2), fetch = !DBG || !(a.ablate & 1);      // (ablation probe: stamped instantiation only)
        if (work && i + 1 < n) s_phase(s_base(i + 1), sreg);      // MFMAs first: they execute while the owner lanes walk the softmax chain
        XSTAMP(1);
        if (work && owner) softmax(i);
        XSTAMP(2);
        __syncthreads();                                  // C: P^T(i) and the rescale factors are in LDS; partials(i) are consumed
        XSTAMP(3);
        if (work && i + 1 < n) write_partials(sreg);
        XSTAMP(4);
        if (work) pv(t_base(i));
        XSTAMP(5);
        if (fetch && i + 2 < n) wait_tile(i + 2);
        XSTAMP(6);
        __syncthreads();                                  // D: partials(i + 1) in LDS; tile i + 2 landed; everybody is done with tile i
        XSTAMP(7);
        const int nx = i + 3 + (1 - half_w);              // waves 4-7: second half of tile i + 3; waves 0-3: first half of tile i + 4
        if (fetch && nx < n) issue(nx);
        XSTAMP(8);
    }
#undef XSTAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XPHASE(12);
    // ---- this split's partial: (m, l) per head, unnormalised O'[head][c] in the order xabs_vup loads B fragments:
    //      part[split][head][c / 8][slot][c & 7]
    if (owner && o_key == 0 && o_head < H) a.ml[((size_t)sp * H + o_head) * a.max_batch + b] = float2{m_run, l_run};
    {
        const int head = lane & 31, hl = lane >> 5;
        if (head < H) {
            float* pb = a.part + (((size_t)sp * H + head) * (D / 8)) * a.max_batch * 8 + (size_t)b * 8 + 4 * hl;
#pragma unroll
            for (int mt = 0; mt < CW; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c8 = (wave * CW + mt) * 4 + g;
                    *reinterpret_cast<float4*>(pb + (size_t)c8 * a.max_batch * 8) = float4{acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
                }
        }
    }
    if constexpr (DBG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XPHASE(13);
  }     // slots of this workgroup
#undef XPHASE
}

// (A register-staged form of this kernel - ordinary buffer loads whose registers are the S-phase A fragments, then a wave-private LDS slot
// for the transpose reads: no encoder byte crosses waves, no LDS-DMA - was built in round 4, is bit-identical and SLOWER at every batch
// size: 63.7 vs 57.2 us at 64 slots, 53.8 vs 45.3 us at 8 (profiles/r04y_*).  The tile period of this kernel is not set by the fetch
// path: about 3 900 cycles per 16-key tile against 830 cycles of MFMA work per SIMD and 3 450 cycles of HBM time at 6.2 TB/s - the rest
// is the lock-step S -> reduce -> softmax -> P -> P V chain between two workgroup barriers.  A role-split form (waves 0-3: S + softmax of
// tile i + 1, waves 4-7: P V of tile i, so that one group's matrix burst runs beside the other's latency chain) is correct as well and
// slower too: 63.2 / 51.4 us (profiles/r04aa_*); so are two re-orderings of the Y interval (profiles/r04z_*).  Code removed; git history
// has all of it.)

// ---------------------------------------------------------------------------------------------- xabs_vup
// att[slot][n] = (W_v[n][:] . sum_s w_s O'_s[head(n)][:]) / l + b_v[n],  w_s = exp(m_s - max m),  l = sum_s w_s l_s.
// A workgroup = one HEAD (the two 32-row tiles of W_v that share its O'_h: the partials are read once) x one K slice x one batch tile;
// W_v is tiled like every decoder projection (Wt[rt][kt][lane][8]); the B fragment of a k tile is the lane's slot, 8 consecutive
// channels: loaded from every split's partial and combined on the fly (index order).  K slices meet through write-through partial
// tiles + a ticket exactly like dec32_proj_kernel.  Every load of a wave is requested before the first use.
// (__launch_bounds__(256, 1): the TW x (2 + 2 S) 16-byte loads of a lane live in registers at once - 200 at d = 1280 - beside 64 accumulators)
template <int S, int TW>
__global__ __launch_bounds__(256, 1) void xabs_vup_kernel(const XabsArgs a, int ks, int n_bt) {
    __shared__ float red[4][2][16][64];
    __shared__ int last_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = a.d, H = a.n_head, KT = d >> 4;
    const int grp8 = blockIdx.x / (8 * n_bt), in8 = blockIdx.x % (8 * n_bt);
    const int xw = grp8 * 8 + (in8 & 7), bt = in8 >> 3;
    if (xw >= H * ks) return;
    const int h = xw % H, ksi = xw / H;
    const int kt0 = (ksi * 4 + wave) * TW;
    const int j = lane & 31, hl = lane >> 5;
    const int gbl = min(bt * 32 + j, a.max_batch - 1);          // fragment loads of padding lanes stay inside the buffers
    float2 ml[S];
#pragma unroll
    for (int s = 0; s < S; ++s) ml[s] = a.ml[((size_t)s * H + h) * a.max_batch + gbl];
    const u32x4* wp = reinterpret_cast<const u32x4*>(a.wv_t) + ((size_t)(2 * h) * KT + kt0) * 64 + lane;
    const size_t sstride = (size_t)H * (d / 8) * a.max_batch * 8;       // floats per split
    const float* pp = a.part + ((size_t)h * (d / 8) + kt0 * 2 + hl) * a.max_batch * 8 + (size_t)gbl * 8;
    u32x4 w0[TW], w1[TW];
    f32x4 p0[TW][S], p1[TW][S];
#pragma unroll
    for (int i = 0; i < TW; ++i) { w0[i] = __builtin_nontemporal_load(wp + (size_t)i * 64); w1[i] = __builtin_nontemporal_load(wp + ((size_t)KT + i) * 64); }
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float* q = pp + (size_t)s * sstride + (size_t)i * 2 * a.max_batch * 8;
            p0[i][s] = *reinterpret_cast<const f32x4*>(q);
            p1[i][s] = *reinterpret_cast<const f32x4*>(q + 4);
        }
    // pin: every request above is issued before anything is consumed (the optimiser otherwise sinks the loads next to their uses:
    // five dependent rounds of memory latency instead of one)
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        asm volatile("" : "+v"(w0[i]), "+v"(w1[i]));
#pragma unroll
        for (int s = 0; s < S; ++s) asm volatile("" : "+v"(p0[i][s]), "+v"(p1[i][s]));
    }
    // ---- split weights of this lane's slot
    float wn[S];
    {
        float mg = ml[0].x;
#pragma unroll
        for (int s = 1; s < S; ++s) mg = fmaxf(mg, ml[s].x);
        float lg = 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s) { wn[s] = __expf(ml[s].x - mg); lg = fmaf(wn[s], ml[s].y, lg); }
        const float inv = 1.0f / lg;
#pragma unroll
        for (int s = 0; s < S; ++s) wn[s] *= inv;
    }
    f32x16 acc_h[2] = {{0}, {0}}, acc_l[2] = {{0}, {0}};
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < S; ++s) {          // index order: the combine fixes the bits
#pragma unroll
            for (int e = 0; e < 4; ++e) { z[e] = fmaf(wn[s], p0[i][s][e], z[e]); z[4 + e] = fmaf(wn[s], p1[i][s][e], z[4 + e]); }
        }
        f16x8 zh, zl;
#pragma unroll
        for (int e = 0; e < 8; ++e) { f16 h_, l_; split_hilo(z[e], h_, l_); zh[e] = h_; zl[e] = l_; }
        const f16x8 wf0 = __builtin_bit_cast(f16x8, w0[i]), wf1 = __builtin_bit_cast(f16x8, w1[i]);
        acc_h[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf0, zh, acc_h[0], 0, 0, 0);
        acc_l[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf0, zl, acc_l[0], 0, 0, 0);
        acc_h[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf1, zh, acc_h[1], 0, 0, 0);
        acc_l[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf1, zl, acc_l[1], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][t][r][lane] = fmaf(acc_l[t][r], 1.0f / 2048.0f, acc_h[t][r]);
    __syncthreads();
    // epilogue coordinates: slot tid & 31, rows 4 sub .. 4 sub + 3 of each of the two row tiles (the accumulator rows 4 wave' + i of half-wave hl)
    const int sub = tid >> 5, gb = bt * 32 + (tid & 31);
    float v[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[t][i] = ((red[0][t][4 * wave + i][lane] + red[1][t][4 * wave + i][lane]) + red[2][t][4 * wave + i][lane]) + red[3][t][4 * wave + i][lane];
    if (ks > 1) {
        float* base = a.kpart + (((size_t)bt * H + h) * ks) * 2048 + tid * 4;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 pv4 = {v[t][0], v[t][1], v[t][2], v[t][3]};
            float* mine = base + (size_t)ksi * 2048 + t * 1024;
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(mine), "v"(pv4) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int* cnt = a.ticket + bt * H + h;
            const int t = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == ks - 1);
            if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = last;
        }
        __syncthreads();
        if (!last_flag) return;
        float pv[4][2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float* p = base + (size_t)(s < ks ? s : 0) * 2048 + t * 1024;
#pragma unroll
                for (int i = 0; i < 4; ++i) pv[s][t][i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = pv[0][t][i];
#pragma unroll
                for (int s = 1; s < 4; ++s) x += (s < ks) ? pv[s][t][i] : 0.0f;
                v[t][i] = x;
            }
    }
    if (gb < a.batch && slot_live(a.seq + gb)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int nn = (2 * h + t) * 32 + 4 * sub;
            const float4 bv = *reinterpret_cast<const float4*>(a.bv + nn);
            const float y[4] = {v[t][0] + bv.x, v[t][1] + bv.y, v[t][2] + bv.z, v[t][3] + bv.w};
            f16x4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) { f16 h_, l_; split_hilo(y[i], h_, l_); hi[i] = h_; lo[i] = l_; }
            const size_t o = plane_index(gb, nn, d);
            *reinterpret_cast<f16x4*>(a.att_hi + o) = hi;
            *reinterpret_cast<f16x4*>(a.att_lo + o) = lo;
        }
    }
}

// ---------------------------------------------------------------------------------------------- launchers
static int xabs_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

bool xabs_supported(int d, int n_head) { return d % 256 == 0 && d >= 512 && d <= 1280 && n_head <= 32; }

// Automatic key splits per slot (a constant of the session): as many as keep slots x splits workgroups within ONE round of the chip's 256 CUs, at most kXabsSplits.
// Up to round 6 the choice was 4 whatever the batch; one large-v3 session alone, ms per decoder step at 4 / 3 / 2 / 1 splits (profiles/r06ah_lone_session_key_splits.jsonl):
// 32 slots 4.21 / 4.52 / 5.16 / 7.28; 96 slots 6.45 / 7.08 / 5.95 / 7.93; 128 slots 7.64 / 7.97 / 6.86 / 8.72; 192 slots 9.84 / 10.52 / 10.04 / 9.55; 256 slots 12.82 / 12.09 / 11.45 / 10.92 -
// a second round of workgroups pays the kernel's exposed prologue and epilogue again.  (Slots that share an encoder output - beam search - are the exception: their streams
// are L2 hits and more workgroups win, 240 audio-s/s with 4 splits against 228 with 2 on configs[4]; such a caller asks for 4: wh_session_create_tuned.)
int xabs_auto_splits(int max_batch) {
    const int s = 256 / (max_batch > 0 ? max_batch : 1);
    return s < 1 ? 1 : (s > kXabsSplits ? kXabsSplits : s);
}
int xabs_splits(int max_batch) {
    const int e = xabs_env("WH_XABS_SPLITS", 0);
    if (e >= 1 && e <= kXabsSplits) return e;
    return xabs_auto_splits(max_batch);
}

void launch_xabs_qk(const XabsArgs& a, int n_bt, hipStream_t st) {
    ProfScope ps_(KK_DEC_XQK, st);
    const dim3 grid(a.d / 256, a.n_head, n_bt);
    if (a.n_head > 16) xabs_qk_kernel<2><<<grid, 256, 0, st>>>(a);
    else xabs_qk_kernel<1><<<grid, 256, 0, st>>>(a);
}

template <int CW, int NHT, bool DBG, bool NTL>
static void launch_attn_k(const XabsArgs& a, hipStream_t st) {
    constexpr int lds = xabs_lds_bytes(CW);
    static PerDeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xabs_attn_kernel<CW, NHT, DBG, NTL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    const int spw = a.spw > 1 ? a.spw : 1, b1 = (a.batch + spw - 1) / spw;
    const int n_grp = (b1 + 3) / 4 * a.n_split;             // (split, 4 slots) groups, 8 of them (one per XCD) to every 32 workgroup ids
    xabs_attn_kernel<CW, NHT, DBG, NTL><<<dim3((unsigned)((n_grp + 7) / 8 * 32)), 512, lds, st>>>(a);
}
template <int CW, int NHT>
static void launch_attn_t(const XabsArgs& a, hipStream_t st) {
    static const int nt = xabs_env("WH_XABS_NT", 1);          // non-temporal policy on the encoder-output stream (in flight: 19.2 k vs 18.1 k sequence-steps/s, profiles/r04l_*); 0 = A/B side
    static const int ablate = xabs_env("WH_XABS_ABLATE", 0);  // timing probe (garbage results): 1 no LDS-DMA in the loop, 2 no S / softmax / P V work, 3 both
    if (a.dbg || ablate) { XabsArgs b = a; b.ablate = ablate; launch_attn_k<CW, NHT, true, false>(b, st); return; }      // the stamped instantiation (tools/xabs_timeline.py)
    // beam search (cross_div > 1: the beams of an audio read ONE encoder output): cacheable loads - the workgroups of an audio's beams are
    // dispatched back to back onto one XCD (4 consecutive slots per group) and the later ones are meant to hit the first one's lines in its L2
    if (nt && a.cross_div <= 1) launch_attn_k<CW, NHT, false, true>(a, st); else launch_attn_k<CW, NHT, false, false>(a, st);
}
void launch_xabs_attn(const XabsArgs& a, hipStream_t st) {
    ProfScope ps_(KK_DEC_CROSS_ATTN, st);
    const bool two = a.n_head > 16;
    switch (a.d / 256) {
        case 2: two ? launch_attn_t<2, 2>(a, st) : launch_attn_t<2, 1>(a, st); break;
        case 3: two ? launch_attn_t<3, 2>(a, st) : launch_attn_t<3, 1>(a, st); break;
        case 4: two ? launch_attn_t<4, 2>(a, st) : launch_attn_t<4, 1>(a, st); break;
        default: two ? launch_attn_t<5, 2>(a, st) : launch_attn_t<5, 1>(a, st); break;
    }
}

void launch_xabs_vup(const XabsArgs& a, int n_bt, hipStream_t st) {
    ProfScope ps_(KK_DEC_XVUP, st);
    // K slices: d / 16 k tiles over 4 waves x ks workgroups, TW tiles per wave; ks = 4 at every supported width (TW = d / 256)
    constexpr int ks = 4;
    const int nx = a.n_head * ks;
    const unsigned grid = (unsigned)(((nx + 7) / 8) * 8 * n_bt);
#define XVUP(S_) do { switch (a.d / 256) { \
        case 2: xabs_vup_kernel<S_, 2><<<grid, 256, 0, st>>>(a, ks, n_bt); break; \
        case 3: xabs_vup_kernel<S_, 3><<<grid, 256, 0, st>>>(a, ks, n_bt); break; \
        case 4: xabs_vup_kernel<S_, 4><<<grid, 256, 0, st>>>(a, ks, n_bt); break; \
        default: xabs_vup_kernel<S_, 5><<<grid, 256, 0, st>>>(a, ks, n_bt); break; } } while (0)
    switch (a.n_split) {
        case 1: XVUP(1); break;
        case 2: XVUP(2); break;
        case 3: XVUP(3); break;
        default: XVUP(4); break;
    }
#undef XVUP
}

}  // namespace wh
