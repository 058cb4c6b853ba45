// Fused projection + attention launches of the decoder step (gfx950): the two places of a layer where the consumer of a projection
// needs only a FEW of its row tiles, not all of them, so the grid-wide kernel boundary between the two can be replaced by an in-launch
// hand-off per (batch tile, head) - and the consumer's own memory stream starts before the projection has finished:
//
//   QKV projection -> self-attention     head h of a batch tile needs the 6 row tiles q / k / v[64 h .. 64 h + 63]; the attention
//                                        workgroups (head, slot) fetch the cached rows < token_index while the projection still
//                                        streams its weights, then read this step's row and the query (6 arrivals)
//   cross query    -> cross-attention    head h needs the 2 row tiles q[64 h ..]; the attention workgroups (key split, head, slot)
//                                        have their K / V rows in flight (the 492 MB stream of the step's dominant kernel) while
//                                        the 3 MB query projection runs - it disappears under that stream (2 arrivals)
//
// One grid, two roles: the first n_prod workgroups are the projection (dec32_body.h, the same code and bits as the stand-alone
// kernel), the rest the attention (dec_attn_body.h).  Workgroups are dispatched in id order per XCD, so every producer is resident
// or done before a consumer of the same launch can occupy a slot in its XCD; consumers poll with a bounded spin (dec_attn_body.h).
// The arrival counters are re-armed by workgroup 0 of the NEXT kernel of the chain (the out projection), after the boundary.
// What this replaces: two of the eight kernel boundaries of a layer of the per-token decoder call
// (Sources/WhisperKit/Core/TextDecoder.swift:381-418).  MI355X_MICROARCH.md prices: "boundary" 1.7-1.9 us + the consumer's
// first-byte latency per cut, against "handoff-flag" 1.3-2.2 us overlapped with the consumer's prefetch ("prefetch-credit").
#include <cstdlib>

#include "dec32_body.h"
#include "dec_attn_body.h"

namespace wh {

template <int TC>
__global__ __launch_bounds__(256, 2) void dec_qkv_self_kernel(const P32Args pa, const AttnArgs at, const int n_prod) {
    if ((int)blockIdx.x < n_prod) { dec32_proj_body<P32_QKV, true, TC>(pa, (int)blockIdx.x); return; }
    const int c = (int)blockIdx.x - n_prod;           // (head, slot), head fastest: the readers of a slot's cache rows sit together
    dec_self_attn_body<7, true>(at, c % at.n_head, c / at.n_head);
}

// (256, 4): at most 128 registers - the attention role keeps 4 workgroups per CU (48 - 64 KB of K / V in flight each)
template <int PASSES>
__global__ __launch_bounds__(256, 4) void dec_cq_cross_kernel(const P32Args pa, const AttnArgs at, const int n_prod) {
    if ((int)blockIdx.x < n_prod) { dec32_proj_body<P32_Q, true, 2>(pa, (int)blockIdx.x); return; }
    const int c = (int)blockIdx.x - n_prod;
    const int S = at.n_split, sp = c % S, hb = c / S;
    dec_cross_attn_body<PASSES, true, true>(at, sp, hb % at.n_head, hb / at.n_head, (unsigned)c);
}

static unsigned producer_grid(const P32Args& a) { return (unsigned)(((((a.N + 31) / 32) * a.ks + 7) / 8) * 8 * a.n_bt); }

unsigned long long* debug_buffer();

// QKV projection + self-attention in one launch.  `a` as for launch_dec32_proj(P32_QKV, ...) with a.signal set; at.ready = a.signal.
bool launch_qkv_self_fused(const P32Args& a_in, const AttnArgs& at_in, int n_bt, int H, int B, hipStream_t st) {
    P32Args a = a_in;
    a.dbg = nullptr;
    a.ks = 1; a.tw = a.K / 64; a.n_bt = n_bt;
    AttnArgs at = at_in;
    at.ready = a.signal; at.ready_need = 6;
    const unsigned n_prod = producer_grid(a), grid = n_prod + (unsigned)(H * B);
    ProfScope ps_(KK_DEC_QKV, st);
    const int tw = a.tw;
    if (tw % 5 == 0) dec_qkv_self_kernel<5><<<grid, 256, 0, st>>>(a, at, (int)n_prod);
    else if (tw % 4 == 0) dec_qkv_self_kernel<4><<<grid, 256, 0, st>>>(a, at, (int)n_prod);
    else if (tw % 3 == 0) dec_qkv_self_kernel<3><<<grid, 256, 0, st>>>(a, at, (int)n_prod);
    else if (tw % 2 == 0) dec_qkv_self_kernel<2><<<grid, 256, 0, st>>>(a, at, (int)n_prod);
    else dec_qkv_self_kernel<1><<<grid, 256, 0, st>>>(a, at, (int)n_prod);
    return true;
}

// cross-query projection + cross-attention in one launch (the projection streams in chunks of 2 k-tiles: K / 64 must be even,
// which holds for every Whisper width; otherwise the caller keeps the two stand-alone kernels)
bool launch_cq_cross_fused(const P32Args& a_in, const AttnArgs& at_in, int n_bt, int S, int H, int B, hipStream_t st) {
    if ((a_in.K / 64) % 2) return false;
    P32Args a = a_in;
    a.dbg = nullptr;
    a.ks = 1; a.tw = a.K / 64; a.n_bt = n_bt;
    AttnArgs at = at_in;
    at.ready = a.signal; at.ready_need = 2; at.no_fence = 1;
    at.dbg = nullptr;
    const unsigned n_prod = producer_grid(a), grid = n_prod + (unsigned)(S * H * B);
    ProfScope ps_(KK_DEC_CROSS_ATTN, st);
    if (S == 6) dec_cq_cross_kernel<8><<<grid, 256, 0, st>>>(a, at, (int)n_prod);
    else if (S == 8) dec_cq_cross_kernel<6><<<grid, 256, 0, st>>>(a, at, (int)n_prod);
    else if (S == 12) dec_cq_cross_kernel<4><<<grid, 256, 0, st>>>(a, at, (int)n_prod);
    else if (S == 24) dec_cq_cross_kernel<2><<<grid, 256, 0, st>>>(a, at, (int)n_prod);
    else return false;
    return true;
}

}  // namespace wh
