// Beam search at temperature 0: BeamSearchTokenSampler + the token loop that drives it.
//
// NO REFERENCE BEHAVIOUR.  The reference declares the sampler (Core/Text/TokenSampler.swift:254-290: beamSize, eotToken, patience,
// maxCandidates = Int(Float(beamSize) * patience), finishedSequences) but its update / finalize are `fatalError("Not implemented")`,
// and TokenSampling.update sees ONE sequence, so a beam cannot even be expressed through that protocol.  BASELINE configs[4] and
// SURVEY 8(d) c5 still ask for "beam=5 per openai/whisper semantics ... labelled no reference behaviour": what is restated here is
// openai/whisper's BeamSearchDecoder (whisper/decoding.py, v20231117: update :343-404, finalize :406-424, MaximumLikelihoodRanker
// :236-255 with length_penalty None) on the reference's decodeText skeleton (prompt pre-fill, logits filters, result conventions).
// The CPU oracle of the same name (oracle/decode.py) is the checker; parity with openai/whisper itself is unpinned (not in the image).
//
// Device side: the ordinary decoder step over n_audio x beam_size slots, the batched LogitsFilter kernel, beam_topk_kernel
// (log-softmax + the beam_size + 1 best entries per slot) and slot copies for the cache rearrangement.  The candidate ranking is host
// code: 30 candidates per audio and step.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "internal.h"

using namespace wh;
using whi::set_error;

namespace whi {

struct BeamSeq {
    std::vector<int> tok;
    std::vector<float> lp;
    float sum = 0.0f;
};

struct BeamSampler {
    int beam_size = 1, eot = 0, max_candidates = 1;
    float patience = 1.0f;
    std::vector<BeamSeq> finished;      // insertion-ordered, like the Python dict of finished sequences

    void reset() { finished.clear(); }

    // decoding.py:343-404 for one audio.  topk_*: per beam the beam_size + 1 best (log-prob, token) pairs, best first.
    bool update(const std::vector<BeamSeq>& beams, const float* topk_lp, const int* topk_tok, int stride, std::vector<BeamSeq>& next,
                std::vector<int>& sources) {
        struct Cand { int rep, tok, src; float score, lp; };
        const int nb = (int)beams.size(), K = beam_size + 1;
        // `scores[sequence] = ...` is keyed by the whole sequence: beams with equal token lists (the first step: every beam is the
        // prompt) share their keys, a later assignment overwrites the value and keeps the key's position
        std::vector<int> rep(nb);
        for (int j = 0; j < nb; ++j) {
            rep[j] = j;
            for (int i = 0; i < j; ++i) if (beams[i].tok == beams[j].tok) { rep[j] = i; break; }
        }
        std::vector<Cand> cands;
        cands.reserve((size_t)nb * K);
        for (int j = 0; j < nb; ++j)
            for (int c = 0; c < K; ++c) {
                const int t = topk_tok[j * stride + c];
                const float lp = topk_lp[j * stride + c];
                const float sc = beams[j].sum + lp;
                bool found = false;
                for (Cand& e : cands) if (e.rep == rep[j] && e.tok == t) { e.src = j; e.score = sc; e.lp = lp; found = true; break; }
                if (!found) cands.push_back(Cand{rep[j], t, j, sc, lp});
            }
        std::stable_sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.score > b.score; });
        next.clear();
        sources.clear();
        std::vector<BeamSeq> newly;
        for (const Cand& e : cands) {
            BeamSeq q;
            q.tok = beams[e.src].tok; q.tok.push_back(e.tok);
            q.lp = beams[e.src].lp; q.lp.push_back(e.lp);
            q.sum = e.score;
            if (e.tok == eot) newly.push_back(std::move(q));
            else {
                next.push_back(std::move(q));
                sources.push_back(e.src);
                if ((int)next.size() == beam_size) break;
            }
        }
        std::stable_sort(newly.begin(), newly.end(), [](const BeamSeq& a, const BeamSeq& b) { return a.sum > b.sum; });
        for (BeamSeq& q : newly) {
            if ((int)finished.size() >= max_candidates) break;      // the candidate list is full
            finished.push_back(std::move(q));
        }
        return (int)finished.size() >= max_candidates;
    }

    // decoding.py:406-424: when fewer than beam_size sequences finished, the live beams follow (best sum first, ties: higher index)
    void finalize(const std::vector<BeamSeq>& beams) {
        if ((int)finished.size() >= beam_size) return;
        std::vector<int> order(beams.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return beams[a].sum < beams[b].sum; });
        for (int k = (int)order.size() - 1; k >= 0; --k) {
            BeamSeq q = beams[order[k]];
            q.tok.push_back(eot);
            q.lp.push_back(0.0f);
            finished.push_back(std::move(q));
            if ((int)finished.size() >= beam_size) break;
        }
    }

    // MaximumLikelihoodRanker, length_penalty None: sum of log-probs / number of sampled tokens before EOT; first maximum
    int rank(int sample_begin) const {
        int best = 0;
        float best_score = -INFINITY;
        for (size_t i = 0; i < finished.size(); ++i) {
            const std::vector<int>& t = finished[i].tok;
            int len = 0;
            for (size_t k = (size_t)sample_begin; k < t.size() && t[k] != eot; ++k) ++len;
            const float sc = finished[i].sum / (float)std::max(len, 1);
            if (sc > best_score) { best = (int)i; best_score = sc; }
        }
        return best;
    }
};

int upload_sampler_cfg(wh_session* s, const wh_decoding_options* opt, const wh_special_tokens* st, int prefilled_index,
                       int initial_prompt_index, int language_filter, uint64_t seed);
}  // namespace whi

struct wh_beam_sampler {
    whi::BeamSampler b;
};

extern "C" int wh_beam_sampler_create(int beam_size, int32_t eot_token, float patience, wh_beam_sampler** out) {
    if (!out) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_beam_sampler_create: null output");
    const int max_candidates = (int)((float)beam_size * patience);                          // TokenSampler.swift:269
    if (beam_size <= 0 || beam_size + 1 > kBeamTopK || max_candidates <= 0)
        return set_error(WH_ERR_INVALID_ARGUMENT, "Invalid beam size %d or patience %g (beam sizes 1..%d)", beam_size, patience, kBeamTopK - 1);   // fatalError in the reference (:273)
    wh_beam_sampler* h = new (std::nothrow) wh_beam_sampler();
    if (!h) return set_error(WH_ERR_OUT_OF_MEMORY, "wh_beam_sampler_create: out of host memory");
    h->b.beam_size = beam_size; h->b.eot = eot_token; h->b.patience = patience; h->b.max_candidates = max_candidates;
    *out = h;
    return WH_OK;
}
extern "C" void wh_beam_sampler_destroy(wh_beam_sampler* h) { delete h; }
extern "C" void wh_beam_sampler_reset(wh_beam_sampler* h) { if (h) h->b.reset(); }
extern "C" int wh_beam_sampler_max_candidates(const wh_beam_sampler* h) { return h ? h->b.max_candidates : -1; }
extern "C" int wh_beam_sampler_finished_count(const wh_beam_sampler* h) { return h ? (int)h->b.finished.size() : -1; }

static void unpack_beams(int n_beams, int len, const int32_t* tokens, const float* token_logprobs, const float* sums, std::vector<whi::BeamSeq>& beams) {
    beams.resize(n_beams);
    for (int j = 0; j < n_beams; ++j) {
        beams[j].tok.assign(tokens + (size_t)j * len, tokens + (size_t)(j + 1) * len);
        if (token_logprobs) beams[j].lp.assign(token_logprobs + (size_t)j * len, token_logprobs + (size_t)(j + 1) * len);
        else beams[j].lp.assign(len, 0.0f);
        beams[j].sum = sums[j];
    }
}

static int wh_beam_sampler_update_impl(wh_beam_sampler* h, int n_beams, int len, const int32_t* tokens, const float* token_logprobs, const float* sums,
                                      const float* topk_logprobs, const int32_t* topk_tokens, int topk_stride, int32_t* new_tokens,
                                      float* new_token_logprobs, float* new_sums, int32_t* sources, int32_t* n_new, int32_t* completed) {
    if (!h || !tokens || !sums || !topk_logprobs || !topk_tokens || !new_tokens || !new_sums || !sources || !n_new || !completed || n_beams < 1 || len < 1 ||
        topk_stride < h->b.beam_size + 1)
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_beam_sampler_update: invalid argument");
    std::vector<whi::BeamSeq> beams, next;
    unpack_beams(n_beams, len, tokens, token_logprobs, sums, beams);
    std::vector<int> src;
    const bool done = h->b.update(beams, topk_logprobs, topk_tokens, topk_stride, next, src);
    *n_new = (int)next.size();
    *completed = done ? 1 : 0;
    for (size_t j = 0; j < next.size(); ++j) {
        memcpy(new_tokens + j * (len + 1), next[j].tok.data(), sizeof(int32_t) * (len + 1));
        if (new_token_logprobs) memcpy(new_token_logprobs + j * (len + 1), next[j].lp.data(), sizeof(float) * (len + 1));
        new_sums[j] = next[j].sum;
        sources[j] = src[j];
    }
    return WH_OK;
}

static int wh_beam_sampler_finalize_impl(wh_beam_sampler* h, int n_beams, int len, const int32_t* tokens, const float* token_logprobs, const float* sums,
                                        int sample_begin, int capacity, int32_t* best_tokens, float* best_token_logprobs, int32_t* best_len,
                                        float* best_sum, int32_t* n_finished) {
    if (!h || !tokens || !sums || !best_tokens || !best_len || n_beams < 1 || len < 1 || sample_begin < 0)
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_beam_sampler_finalize: invalid argument");
    std::vector<whi::BeamSeq> beams;
    unpack_beams(n_beams, len, tokens, token_logprobs, sums, beams);
    h->b.finalize(beams);
    const whi::BeamSeq& q = h->b.finished[h->b.rank(sample_begin)];
    if ((int)q.tok.size() > capacity) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_beam_sampler_finalize: capacity %d < %d tokens", capacity, (int)q.tok.size());
    memcpy(best_tokens, q.tok.data(), sizeof(int32_t) * q.tok.size());
    if (best_token_logprobs) memcpy(best_token_logprobs, q.lp.data(), sizeof(float) * q.lp.size());
    *best_len = (int)q.tok.size();
    if (best_sum) *best_sum = q.sum;
    if (n_finished) *n_finished = (int)h->b.finished.size();
    return WH_OK;
}

// ------------------------------------------------------------------------------------------------ the token loop
namespace {
struct AudioBeams {
    whi::BeamSampler sampler;
    std::vector<whi::BeamSeq> beams;
    bool live = false;          // still expanding
    bool early = false;         // the pre-fill ended the window (EOT / first-token threshold): the greedy state is the result
    int steps = 0;
    int first_token_too_low = 0;
};

int ensure_beam_buffers(wh_session* s) {
    if (!s->beam_owner) WH_HIP(hipMalloc((void**)&s->beam_owner, sizeof(int) * kMaxTok * (size_t)s->B));
    if (!s->beam_lp) WH_HIP(hipMalloc((void**)&s->beam_lp, sizeof(float) * kBeamTopK * (size_t)s->B));
    if (!s->beam_tok) WH_HIP(hipMalloc((void**)&s->beam_tok, sizeof(int) * kBeamTopK * (size_t)s->B));
    return WH_OK;
}
}  // namespace

static int wh_decode_text_beam_impl(wh_session* s, int n_audio, int beam_size, float patience, const wh_decoding_options* opt,
                                   const wh_special_tokens* st, const int32_t* prompt, int n_prompt, const int32_t* language_tokens,
                                   wh_decoding_result* out) {
    CHECK_SESSION(s);       // selects the session's device: the beam buffers below are allocated on first use, from any host thread
    if (!opt || !st || !prompt || !out) return set_error(WH_ERR_DECODING_FAILED, "wh_decode_text_beam: null argument");
    const int max_candidates = (int)((float)beam_size * patience);
    if (beam_size <= 0 || beam_size + 1 > kBeamTopK || max_candidates <= 0)
        return set_error(WH_ERR_INVALID_ARGUMENT, "Invalid beam size %d or patience %g (beam sizes 1..%d)", beam_size, patience, kBeamTopK - 1);
    if (n_audio < 1 || (long long)n_audio * beam_size > s->B)
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_decode_text_beam: %d audios x %d beams need %d slots, the session has %d", n_audio, beam_size,
                         n_audio * beam_size, s->B);
    if (n_prompt < 1 || n_prompt >= kMaxTok) return set_error(WH_ERR_PREFILL_FAILED, "wh_decode_text_beam: prompt length %d out of range [1,%d)", n_prompt, kMaxTok);
    if (opt->word_timestamps) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_decode_text_beam: word timestamps are not recorded along beams (decode the chosen tokens again with wh_decode_text and prefixTokens to align them)");
    const int V = s->m->dims.n_vocab, H = s->m->dims.n_text_head;
    (void)H;
    for (int i = 0; i < n_prompt; ++i)
        if (prompt[i] < 0 || prompt[i] >= V) return set_error(WH_ERR_PREFILL_FAILED, "wh_decode_text_beam: prompt token %d out of vocabulary", prompt[i]);
    int r = ensure_beam_buffers(s);
    if (r) return r;
    const int prefilled_index = 0;
    r = whi::upload_sampler_cfg(s, opt, st, prefilled_index, n_prompt, 0, 0);
    if (r) return r;
    s->align_enabled = false;
    const int loop_count = std::max(std::min(opt->sample_length, kMaxTok - 1), 0);
    int lang_pos = -1;
    if (language_tokens && wh_is_model_multilingual(s->m))
        for (int i = 0; i + 1 < n_prompt; ++i) if (prompt[i] == st->start_of_transcript_token) { lang_pos = i + 1; break; }

    // ---- 1. pre-fill on the audios' own slots: decodeText's greedy steps for tokenIndex < n_prompt - 1 (TextDecoder.swift:573-757 on the
    // device state machine), which also resolve the "last prompt timestamp is replaced by the prediction" rule for position n_prompt - 1
    for (int b = 0; b < n_audio; ++b) {
        SeqState& q = s->seq_host[b];
        memset(&q, 0, sizeof(q));
        for (int i = 0; i < n_prompt; ++i) q.tokens[i] = prompt[i];
        if (lang_pos >= 0 && language_tokens[b] >= 0 && language_tokens[b] < V) q.tokens[lang_pos] = language_tokens[b];
        q.n_tokens = n_prompt; q.token_index = prefilled_index; q.next_token = q.tokens[0]; q.prompt_len = n_prompt; q.active = 1; q.temperature = 0.0f;
    }
    s->fused_greedy = false;
    WH_HIP(hipMemcpyAsync(s->seq, s->seq_host, sizeof(SeqState) * n_audio, hipMemcpyHostToDevice, s->st));
    launch_rules_init(s->cfg_dev, s->seq, n_audio, s->st);
    const int n_prefill = std::min(n_prompt - 1, loop_count);
    for (int step = 0; step < n_prefill; ++step) {
        DecodeBuffers db = whi::decode_buffers(s, n_audio, step);
        launch_decoder_step(db, s->cfg_dev, s->suppress_dev, true, s->st);
        WH_CHECK_LAUNCH();
    }
    WH_HIP(hipMemcpyAsync(s->seq_host, s->seq, sizeof(SeqState) * n_audio, hipMemcpyDeviceToHost, s->st));
    WH_HIP(hipStreamSynchronize(s->st));
    std::vector<AudioBeams> A(n_audio);
    std::vector<SeqState> greedy(s->seq_host, s->seq_host + n_audio);
    bool any_live = false;
    for (int a = 0; a < n_audio; ++a) {
        AudioBeams& ab = A[a];
        ab.sampler.beam_size = beam_size; ab.sampler.eot = st->end_token; ab.sampler.patience = patience; ab.sampler.max_candidates = max_candidates;
        ab.steps = greedy[a].steps;
        ab.first_token_too_low = greedy[a].first_token_too_low;
        ab.early = greedy[a].done != 0 || n_prompt - 1 >= loop_count;
        ab.live = !ab.early;
        if (ab.live) {
            whi::BeamSeq q;
            q.tok.assign(greedy[a].tokens, greedy[a].tokens + greedy[a].n_tokens);      // the prompt, its last timestamp resolved
            q.lp.assign(q.tok.size(), 0.0f);
            ab.beams.assign(beam_size, q);
            any_live = true;
        }
    }
    // ---- 2. nothing is replicated.  Audio a owns slots a * beam .. a * beam + beam - 1; the cross K / V is read from the one copy in slot a
    // (DecodeBuffers.cross_div = beam_size; round 2 copied 245 MB per beam at large-v3 and streamed it 5 x per step), and the self-attention
    // cache is read through a row -> owner table (DecodeBuffers.self_owner): the pre-filled rows 0 .. n_prompt - 2 of audio a live in slot a
    // (where the pre-fill wrote them; the beam that occupies slot a only ever writes rows >= n_prompt - 1), row r of a beam's history lives
    // in the slot of the beam that computed it.  Re-parenting a beam copies ints on the host (round 2 / early round 3: openai's
    // rearrange_kv_cache as slot copies through a scratch cache - 18 % of the beam pass at large-v3, profiles/r03i_beam_*).
    const int n_slots = n_audio * beam_size;
    std::vector<int> owner((size_t)n_slots * kMaxTok), owner_next;
    for (int j = 0; j < n_slots; ++j)
        for (int r_ = 0; r_ < kMaxTok; ++r_) owner[(size_t)j * kMaxTok + r_] = r_ < n_prompt - 1 ? j / beam_size : j;
    // ---- 3. the beam loop (decoding.py _main_loop with the reference's loop bounds)
    std::vector<float> h_lp((size_t)n_slots * kBeamTopK);
    std::vector<int> h_tok((size_t)n_slots * kBeamTopK);
    for (int token_index = n_prompt - 1; any_live && token_index < loop_count; ++token_index) {
        if (s->cancel_flag && *s->cancel_flag) { hipStreamSynchronize(s->st); return set_error(WH_ERR_CANCELLED, "wh_decode_text_beam: cancelled through the session's cancel flag"); }
        for (int a = 0; a < n_audio; ++a)
            for (int j = 0; j < beam_size; ++j) {
                SeqState& q = s->seq_host[a * beam_size + j];
                memset(&q, 0, sizeof(q));
                if (!A[a].live || j >= (int)A[a].beams.size()) continue;     // inactive slot: every kernel returns early
                const whi::BeamSeq& bq = A[a].beams[j];
                std::copy(bq.tok.begin(), bq.tok.end(), q.tokens);
                q.n_tokens = (int)bq.tok.size(); q.token_index = token_index; q.next_token = bq.tok.back(); q.prompt_len = n_prompt; q.active = 1;
            }
        WH_HIP(hipMemcpyAsync(s->seq, s->seq_host, sizeof(SeqState) * n_slots, hipMemcpyHostToDevice, s->st));
        for (int j = 0; j < n_slots; ++j) owner[(size_t)j * kMaxTok + token_index] = j;      // the row this step writes is the slot's own
        WH_HIP(hipMemcpyAsync(s->beam_owner, owner.data(), sizeof(int) * owner.size(), hipMemcpyHostToDevice, s->st));
        DecodeBuffers db = whi::decode_buffers(s, n_slots, token_index);
        db.cross_div = beam_size;
        db.self_owner = s->beam_owner;
        launch_decoder_step(db, nullptr, nullptr, false, s->st);
        launch_beam_filter_topk(s->cfg_dev, s->suppress_dev, s->seq, s->logits, n_slots, beam_size + 1, s->beam_lp, s->beam_tok, s->st);
        WH_CHECK_LAUNCH();
        WH_HIP(hipMemcpyAsync(h_lp.data(), s->beam_lp, sizeof(float) * h_lp.size(), hipMemcpyDeviceToHost, s->st));
        WH_HIP(hipMemcpyAsync(h_tok.data(), s->beam_tok, sizeof(int) * h_tok.size(), hipMemcpyDeviceToHost, s->st));
        WH_HIP(hipStreamSynchronize(s->st));      // (also: `owner` / seq_host may be rewritten now)
        std::vector<int> pairs;
        any_live = false;
        for (int a = 0; a < n_audio; ++a) {
            AudioBeams& ab = A[a];
            if (!ab.live) continue;
            ab.steps += 1;
            const float* lp = h_lp.data() + (size_t)a * beam_size * kBeamTopK;
            const int* tk = h_tok.data() + (size_t)a * beam_size * kBeamTopK;
            if (token_index == prefilled_index && !std::isnan(opt->first_token_log_prob_threshold) && lp[0] < opt->first_token_log_prob_threshold) {
                ab.first_token_too_low = 1; ab.live = false; ab.early = true;       // TextDecoder.swift:662-667 on the best first token
                SeqState& g = greedy[a]; g.first_token_too_low = 1;
                continue;
            }
            if ((int)ab.beams[0].tok.size() >= kMaxTok - 1) { ab.live = false; continue; }     // :669 isSegmentCompleted by length
            std::vector<whi::BeamSeq> next;
            std::vector<int> src;
            const bool completed = ab.sampler.update(ab.beams, lp, tk, kBeamTopK, next, src);
            for (int j = 0; j < (int)src.size(); ++j) if (src[j] != j) { pairs.push_back(a * beam_size + src[j]); pairs.push_back(a * beam_size + j); }
            ab.beams = std::move(next);
            if (completed) ab.live = false;
            any_live |= ab.live;
        }
        if (any_live && !pairs.empty()) {
            // rearrange_kv_cache without moving a byte: new beam `to` continues the history of old beam `from` (rows 0 .. token_index)
            owner_next = owner;
            for (size_t p = 0; p < pairs.size(); p += 2)
                std::copy(owner.begin() + (size_t)pairs[p] * kMaxTok, owner.begin() + (size_t)pairs[p] * kMaxTok + token_index + 1,
                          owner_next.begin() + (size_t)pairs[p + 1] * kMaxTok);
            owner.swap(owner_next);
        }
    }
    // ---- 4. finalize, rank, results in the reference's DecodingResult conventions (TextDecoder.swift:776-854)
    for (int a = 0; a < n_audio; ++a) {
        AudioBeams& ab = A[a];
        SeqState fin = greedy[a];
        if (!ab.early) {
            ab.sampler.finalize(ab.beams);
            const whi::BeamSeq& q = ab.sampler.finished[ab.sampler.rank(n_prompt)];
            const int n = std::min((int)q.tok.size(), kMaxTok + 8);
            std::copy(q.tok.begin(), q.tok.begin() + n, fin.tokens);
            std::copy(q.lp.begin(), q.lp.begin() + n, fin.logprobs);
            fin.n_tokens = n;
        }
        fin.steps = ab.steps;
        fin.first_token_too_low = ab.first_token_too_low;
        whi::finalize_decoding_result(fin, opt, st, 0.0f, &out[a]);
    }
    return WH_OK;
}

// ---- C-ABI shims: the bodies above allocate std::vectors (per token inside the beam loop); no exception leaves the library
extern "C" int wh_beam_sampler_update(wh_beam_sampler* h, int n_beams, int len, const int32_t* tokens, const float* token_logprobs, const float* sums,
                                      const float* topk_logprobs, const int32_t* topk_tokens, int topk_stride, int32_t* new_tokens,
                                      float* new_token_logprobs, float* new_sums, int32_t* sources, int32_t* n_new, int32_t* completed) {
    WH_TRY return wh_beam_sampler_update_impl(h, n_beams, len, tokens, token_logprobs, sums, topk_logprobs, topk_tokens, topk_stride, new_tokens, new_token_logprobs, new_sums, sources, n_new, completed); WH_CATCH("wh_beam_sampler_update")
}
extern "C" int wh_beam_sampler_finalize(wh_beam_sampler* h, int n_beams, int len, const int32_t* tokens, const float* token_logprobs, const float* sums,
                                        int sample_begin, int capacity, int32_t* best_tokens, float* best_token_logprobs, int32_t* best_len,
                                        float* best_sum, int32_t* n_finished) {
    WH_TRY return wh_beam_sampler_finalize_impl(h, n_beams, len, tokens, token_logprobs, sums, sample_begin, capacity, best_tokens, best_token_logprobs, best_len, best_sum, n_finished); WH_CATCH("wh_beam_sampler_finalize")
}
extern "C" int wh_decode_text_beam(wh_session* s, int n_audio, int beam_size, float patience, const wh_decoding_options* opt,
                                   const wh_special_tokens* st, const int32_t* prompt, int n_prompt, const int32_t* language_tokens,
                                   wh_decoding_result* out) {
    WH_TRY return wh_decode_text_beam_impl(s, n_audio, beam_size, patience, opt, st, prompt, n_prompt, language_tokens, out); WH_CATCH("wh_decode_text_beam")
}
