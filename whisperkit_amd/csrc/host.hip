// Host-side restatement (C++) of the reference's Swift orchestration around the model stages:
//   TextDecoder.decodeText / detectLanguage / prefillDecoderInputs  (Core/TextDecoder.swift)
//   TranscribeTask.run / decodeWithFallback                        (Core/TranscribeTask.swift)
//   SegmentSeeker.findSeekPointAndSegments / dynamicTimeWarping    (Core/Text/SegmentSeeker.swift)
//   DecodingFallback, TextUtilities.compressionRatio, EnergyVAD, VADAudioChunker, prepareSeekClips
// The token loop itself runs on the device (decoder.hip); this file drives it with replayed hipGraphs
// (one graph = kStepsPerGraph decoder steps, no host round trip inside) and turns the device-side
// SeqState into the reference's DecodingResult / TranscriptionSegment values.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <tuple>

#include "internal.h"

using namespace wh;
using whi::set_error;

namespace whi {
int upload_sampler_cfg(wh_session* s, const wh_decoding_options* opt, const wh_special_tokens* st, int prefilled_index,
                       int initial_prompt_index, int language_filter, uint64_t seed);
}


// CFAbsoluteTimeGetCurrent(): seconds since 2001-01-01 00:00:00 UTC
static inline double cf_absolute_time() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count() - 978307200.0; }
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static inline float f16_round(float v) { return (float)(_Float16)v; }   // FloatType == Float16 on arm64 (ArgmaxCore/FloatType.swift:9-13)

// ------------------------------------------------------------------------------------------------ small utilities
extern "C" float wh_compression_ratio(const int32_t* tokens, int n) {
    // TextUtilities.compressionRatio(of: [Int]): bytes of the Int32 array / bytes of NSData.compressed(using: .zlib)
    // (raw DEFLATE, level 5).  Empty data -> compression throws -> +inf.
    if (!tokens || n <= 0) return INFINITY;
    uLong src_len = (uLong)n * 4;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, 5, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return INFINITY;
    std::vector<unsigned char> outb(deflateBound(&zs, src_len) + 16);
    zs.next_in = (Bytef*)tokens; zs.avail_in = src_len; zs.next_out = outb.data(); zs.avail_out = (uInt)outb.size();
    int rc = deflate(&zs, Z_FINISH);
    uLong clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || clen == 0) return INFINITY;
    return (float)src_len / (float)clen;
}

extern "C" int wh_decoding_fallback(const wh_decoding_options* opt, int first_token_too_low, float no_speech_prob, float compression_ratio,
                                    float avg_logprob, int32_t* needs_fallback) {
    // DecodingFallback.init? - NOTE: order matters (Core/Models.swift:365)
    int reason = WH_FALLBACK_NONE, need = 0;
    if (first_token_too_low) { reason = WH_FALLBACK_FIRST_TOKEN_LOGPROB; need = 1; }
    else if (opt && !isnan(opt->no_speech_threshold) && no_speech_prob > opt->no_speech_threshold) { reason = WH_FALLBACK_SILENCE; need = 0; }
    else if (opt && !isnan(opt->compression_ratio_threshold) && compression_ratio > opt->compression_ratio_threshold) { reason = WH_FALLBACK_COMPRESSION_RATIO; need = 1; }
    else if (opt && !isnan(opt->log_prob_threshold) && avg_logprob < opt->log_prob_threshold) { reason = WH_FALLBACK_LOGPROB; need = 1; }
    if (needs_fallback) *needs_fallback = need;
    return reason;
}

extern "C" int wh_dynamic_time_warping(const float* matrix, int rows, int cols, int32_t* text_idx, int32_t* time_idx, int capacity) {
    // SegmentSeeker.dynamicTimeWarping: cost in Double over -matrix, strict-less tie-breaking (diag, up, else left), backtrace.
    if (!matrix || rows < 1 || cols < 1) return -1;
    const size_t W = (size_t)cols + 1;
    std::vector<double> cost((size_t)(rows + 1) * W, INFINITY);
    std::vector<signed char> trace((size_t)(rows + 1) * W, -1);
    cost[0] = 0;
    for (int j = 1; j <= cols; ++j) trace[j] = 2;
    for (int i = 1; i <= rows; ++i) trace[(size_t)i * W] = 1;
    for (int r = 1; r <= rows; ++r) {
        const double* cp = &cost[(size_t)(r - 1) * W];
        double* cc = &cost[(size_t)r * W];
        signed char* tr = &trace[(size_t)r * W];
        const float* mv = matrix + (size_t)(r - 1) * cols;
        for (int c = 1; c <= cols; ++c) {
            double v = -(double)mv[c - 1];
            double c0 = cp[c - 1] + v, c1 = cp[c] + v, c2 = cc[c - 1] + v;
            if (c0 < c1 && c0 < c2) { cc[c] = c0; tr[c] = 0; }
            else if (c1 < c0 && c1 < c2) { cc[c] = c1; tr[c] = 1; }
            else { cc[c] = c2; tr[c] = 2; }
        }
    }
    std::vector<int> ti, tj;
    int i = rows, j = cols;
    while (i > 0 || j > 0) {
        ti.push_back(i - 1); tj.push_back(j - 1);
        int t = trace[(size_t)i * W + j];
        if (t == 0) { --i; --j; } else if (t == 1) --i; else if (t == 2) --j; else break;
    }
    int n = (int)ti.size();
    if (text_idx && time_idx) {
        if (n > capacity) return -n;
        for (int k = 0; k < n; ++k) { text_idx[k] = ti[n - 1 - k]; time_idx[k] = tj[n - 1 - k]; }
    }
    return n;
}

extern "C" int wh_vad_voice_activity(const float* pcm, int n, int frame_len, int frame_overlap, float thr, uint8_t* out, int capacity) {
    // EnergyVAD.voiceActivity -> AudioProcessor.calculateVoiceActivityInChunks (vDSP_rmsqv per frame > threshold)
    if (n < 0 || frame_len <= 0 || (n > 0 && !pcm)) return -1;
    int count = (int)((n + (long long)frame_len - 1) / frame_len);
    if (!out) return count;
    if (count > capacity) return -count;
    for (int i = 0; i < count; ++i) {
        long long s0 = (long long)i * frame_len, e0 = std::min<long long>(s0 + frame_len + frame_overlap, n);
        double acc = 0;
        for (long long k = s0; k < e0; ++k) acc += (double)pcm[k] * pcm[k];
        float rms = e0 > s0 ? (float)sqrt(acc / (double)(e0 - s0)) : 0.0f;
        out[i] = rms > thr ? 1 : 0;
    }
    return count;
}

static std::vector<std::pair<int, int>> prepare_seek_clips(const wh_decoding_options* opt, int content_frames) {
    // DecodingOptions.prepareSeekClips (Utilities/Extensions+Internal.swift:112-130)
    std::vector<int> pts;
    if (opt && opt->clip_timestamps)
        for (int i = 0; i < opt->n_clip_timestamps; ++i) pts.push_back((int)roundf(opt->clip_timestamps[i] * (float)WH_SAMPLE_RATE));
    if (pts.empty()) pts.push_back(0);
    if (pts.size() % 2 == 1) pts.push_back(content_frames);
    std::vector<std::pair<int, int>> clips;
    for (size_t i = 0; i < pts.size(); i += 2) clips.push_back({pts[i], i + 1 < pts.size() ? pts[i + 1] : content_frames});
    return clips;
}

extern "C" int wh_prepare_seek_clips(const wh_decoding_options* opt, int content_frames, int32_t* clip_start, int32_t* clip_end, int capacity) {
    auto clips = prepare_seek_clips(opt, content_frames);
    if (clip_start && clip_end) {
        if ((int)clips.size() > capacity) return -(int)clips.size();
        for (size_t i = 0; i < clips.size(); ++i) { clip_start[i] = clips[i].first; clip_end[i] = clips[i].second; }
    }
    return (int)clips.size();
}

static bool longest_silence(const std::vector<uint8_t>& v, int* s0, int* e0) {   // VoiceActivityDetector.findLongestSilence
    int best = 0;
    bool found = false;
    size_t i = 0;
    while (i < v.size()) {
        if (v[i]) { ++i; continue; }
        size_t e = i;
        while (e < v.size() && !v[e]) ++e;
        if ((int)(e - i) > best) { best = (int)(e - i); *s0 = (int)i; *e0 = (int)e; found = true; }
        i = e;
    }
    return found;
}

extern "C" int wh_vad_chunk_all(const float* pcm, int n, int max_chunk, const wh_decoding_options* opt, int32_t* cs, int32_t* ce, int capacity) {
    // VADAudioChunker.chunkAll (Core/Audio/AudioChunker.swift:66-107), EnergyVAD() defaults, windowPadding 16000
    if (n < 0 || max_chunk <= 0 || (n > 0 && !pcm)) return -1;
    std::vector<std::pair<int, int>> out;
    const int frame = (int)(0.1f * (float)WH_SAMPLE_RATE), window_padding = 16000;
    if (n <= max_chunk) out.push_back({0, n});
    else {
        for (auto clip : prepare_seek_clips(opt, n)) {
            int start = clip.first;
            while (start < clip.second - window_padding) {
                if (start < 0 || start >= n) return -1;
                int end = clip.second;
                if ((long long)start + max_chunk < end) {
                    int e = std::min(n, start + max_chunk);
                    int mid = start + (e - start) / 2;
                    int cnt = wh_vad_voice_activity(pcm + mid, e - mid, frame, 0, 0.02f, nullptr, 0);
                    std::vector<uint8_t> va(cnt);
                    wh_vad_voice_activity(pcm + mid, e - mid, frame, 0, 0.02f, va.data(), cnt);
                    int s0, e0;
                    if (longest_silence(va, &s0, &e0)) end = mid + (s0 + (e0 - s0) / 2) * frame;
                    else end = e;
                }
                if (!(end > start)) break;
                out.push_back({start, end});
                start = end;
            }
        }
    }
    if (cs && ce) {
        if ((int)out.size() > capacity) return -(int)out.size();
        for (size_t i = 0; i < out.size(); ++i) { cs[i] = out[i].first; ce[i] = out[i].second; }
    }
    return (int)out.size();
}

// ------------------------------------------------------------------------------------------------ prompt
extern "C" int wh_prefill_prompt(const wh_model* m, const wh_decoding_options* opt, const wh_special_tokens* st, int32_t language_token,
                                 int32_t* out, int capacity) {
    // prefillDecoderInputs (Core/TextDecoder.swift:163-216)
    if (!m || !st || !out) return -1;
    std::vector<int> p{st->start_of_transcript_token};
    if (opt) {
        if (wh_is_model_multilingual(m)) {
            p.push_back(language_token >= 0 ? language_token : st->english_token);
            p.push_back(opt->task == 1 ? st->translate_token : st->transcribe_token);
        }
        p.push_back(opt->without_timestamps ? st->no_timestamps_token : st->time_token_begin);
        if (opt->prompt_tokens) {
            const int maxPromptLen = (WH_MAX_TOKEN_CONTEXT / 2) - 1;
            int n = opt->n_prompt_tokens, from = std::max(0, n - maxPromptLen);
            std::vector<int> t{st->start_of_previous_token};
            for (int i = from; i < n; ++i) if (opt->prompt_tokens[i] < st->special_token_begin) t.push_back(opt->prompt_tokens[i]);
            t.insert(t.end(), p.begin(), p.end());
            p.swap(t);
        }
        if (opt->prefix_tokens) {
            int n = opt->n_prefix_tokens, from = std::max(0, n - WH_MAX_TOKEN_CONTEXT / 2);
            for (int i = from; i < n; ++i) if (opt->prefix_tokens[i] < st->special_token_begin) p.push_back(opt->prefix_tokens[i]);
        }
    }
    if ((int)p.size() > capacity) return -(int)p.size();
    for (size_t i = 0; i < p.size(); ++i) out[i] = p[i];
    return (int)p.size();
}

// ------------------------------------------------------------------------------------------------ decodeText
namespace whi {
void finalize_decoding_result(const SeqState& sq, const wh_decoding_options* opt, const wh_special_tokens* st, float temperature,
                              wh_decoding_result* out) {
    // TextDecoder.swift:776-854: finalize (append EOT), slice SOT...EOT, avg log prob, compression ratio, fallback
    memset(out, 0, sizeof(*out));
    std::vector<int> tok(sq.tokens, sq.tokens + sq.n_tokens);
    std::vector<float> lp(sq.logprobs, sq.logprobs + sq.n_tokens);
    if (tok.empty() || tok.back() != st->end_token) { tok.push_back(st->end_token); lp.push_back(0.0f); }
    int start = 0, end = (int)tok.size();
    for (int i = 0; i < (int)tok.size(); ++i) if (tok[i] == st->start_of_transcript_token) { start = i; break; }
    for (int i = 0; i < (int)tok.size(); ++i) if (tok[i] == st->end_token) { end = i; break; }
    if (end < start) end = (int)tok.size() - 1;
    int n = std::min(end - start + 1, WH_MAX_RESULT_TOKENS);
    float sum = 0.0f;
    std::vector<int32_t> words;
    out->language_token = -1;
    for (int i = 0; i < n; ++i) {
        out->tokens[i] = tok[start + i];
        out->token_logprobs[i] = lp[start + i];
        sum += lp[start + i];
        if (tok[start + i] < st->special_token_begin) words.push_back(tok[start + i]);
        if (out->language_token < 0 && tok[start + i] >= st->language_token_begin && tok[start + i] < st->language_token_begin + st->n_language_tokens)
            out->language_token = tok[start + i];
    }
    out->n_tokens = n;
    out->avg_logprob = sum / (float)n;
    out->compression_ratio = wh_compression_ratio(words.data(), (int)words.size());
    out->no_speech_prob = 0.0f;   // TextDecoder.swift:802 (TODO in the reference)
    out->temperature = roundf(f16_round(temperature) * 1000.0f) / 1000.0f;   // Float(sampler.temperature).rounded(3)
    out->is_first_token_logprob_too_low = sq.first_token_too_low;
    out->steps = sq.steps;
    out->fallback_reason = wh_decoding_fallback(opt, sq.first_token_too_low, out->no_speech_prob, out->compression_ratio, out->avg_logprob,
                                                &out->needs_fallback);
}
}  // namespace whi

constexpr int kStepsPerGraph = 8;

static bool use_graphs() {
    static const bool v = [] { const char* e = getenv("WH_NO_GRAPH"); return !(e && e[0] == '1'); }();
    return v;
}

// Step graphs live in the session (a session is driven by one host thread at a time): no process-wide cache, no lock.
// `first_step`: token_index of the live slots at the graph's first step (decodeText starts every slot at 0 and advances them in
// lock step): the graph's self-attention launches fetch only the cache rows its 8 steps can reach, so there is one graph per
// 8 positions (28 per key at most).
static int get_step_graph(wh_session* s, int batch, int first_step, hipGraphExec_t* out) {
    DecodeBuffers db = whi::decode_buffers(s, batch, first_step + kStepsPerGraph - 1);
    const WhGraphKey key{batch, s->align_enabled ? 1 : 0, s->fused_greedy ? 1 : 0, s->align_enabled ? s->n_align_alloc : 0, db.self_rows, db.xattn_gate ? 1 : 0};
    auto it = s->graphs.find(key);
    if (it != s->graphs.end()) { s->graph_use[key] = ++s->graph_tick; *out = it->second; return WH_OK; }
    // The cache is capped (a large-v3 step graph holds ~2.5 k kernel nodes; a configuration = everything of the key but the row bound has up
    // to 28 graphs): when it is full, the configuration that was used longest ago goes - never the one being extended.  The stream is
    // drained first: no executable graph is destroyed while a launch of it may still be running.
    static const size_t cap = [] { const char* e = getenv("WH_GRAPH_CAP"); const int v = e ? atoi(e) : 0; return (size_t)(v > 0 ? v : 4 * 28); }();
    while (s->graphs.size() >= cap) {
        auto same_cfg = [](const WhGraphKey& a, const WhGraphKey& b) {
            return a.batch == b.batch && a.align == b.align && a.fused == b.fused && a.n_align == b.n_align && a.gate == b.gate;
        };
        const WhGraphKey* victim = nullptr;
        unsigned long long victim_last = ~0ull;
        for (const auto& kv : s->graphs) {
            if (same_cfg(kv.first, key)) continue;
            unsigned long long last = 0;                       // a configuration's age = its most recent use
            for (const auto& u : s->graph_use) if (same_cfg(u.first, kv.first)) last = std::max(last, u.second);
            if (last < victim_last) { victim_last = last; victim = &kv.first; }
        }
        if (!victim) break;                                    // only the current configuration is cached: let it grow to its 28
        const WhGraphKey v = *victim;
        WH_HIP(hipStreamSynchronize(s->st));
        for (auto g = s->graphs.begin(); g != s->graphs.end();) {
            if (same_cfg(g->first, v)) { hipGraphExecDestroy(g->second); s->graph_use.erase(g->first); g = s->graphs.erase(g); } else ++g;
        }
    }
    hipGraph_t graph;
    WH_HIP(hipStreamBeginCapture(s->st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < kStepsPerGraph; ++i) launch_decoder_step(db, s->cfg_dev, s->suppress_dev, true, s->st);
    WH_HIP(hipStreamEndCapture(s->st, &graph));
    hipGraphExec_t exec;
    WH_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipGraphDestroy(graph);
    s->graphs[key] = exec;
    s->graph_use[key] = ++s->graph_tick;
    *out = exec;
    return WH_OK;
}

namespace whi {
void drop_session_graphs(wh_session* s) {
    for (auto& kv : s->graphs) hipGraphExecDestroy(kv.second);
    s->graphs.clear();
    s->graph_use.clear();
}
}

static inline bool cancelled(const wh_session* s) { return s->cancel_flag && *s->cancel_flag != 0; }
#define CHECK_CANCEL(s) do { if (cancelled(s)) return set_error(WH_ERR_CANCELLED, "%s: cancelled through the session's cancel flag", __func__); } while (0)

// TranscriptionCallback on a consistent snapshot of the slot states (the stream is idle): one call per unfinished slot; a zero
// return marks the slot done on the device (stream-ordered, before the next step graph) - earlyStopActor semantics.
static int report_progress(wh_session* s, int batch) {
    static const int kOne = 1;
    for (int b = 0; b < batch; ++b) {
        SeqState& q = s->seq_host[b];
        if (!q.active || q.done || q.n_tokens <= 0) continue;
        wh_progress p{};
        p.slot = b; p.n_tokens = q.n_tokens; p.tokens = q.tokens;
        float sum = 0;
        for (int i = 0; i < q.n_tokens; ++i) sum += q.logprobs[i];
        p.avg_logprob = sum / (float)q.n_tokens;
        p.compression_ratio = wh_compression_ratio(q.tokens, q.n_tokens);
        std::string text;
        if (s->tok) {
            std::vector<int> ids;
            for (int i = 0; i < q.n_tokens; ++i) if (!s->skip_special_in_progress || q.tokens[i] < s->special_begin_in_progress) ids.push_back(q.tokens[i]);
            text = s->tok->decode(ids);
            p.text = text.c_str();
        }
        if (!s->progress_cb(s->progress_user, &p)) {
            q.done = 1;
            WH_HIP(hipMemcpyAsync(&s->seq[b].done, &kOne, sizeof(int), hipMemcpyHostToDevice, s->st));
        }
    }
    return WH_OK;
}

static int run_token_loop(wh_session* s, int batch, int loop_count) {
    // every slot's state lives on the device; the host only replays step graphs and polls the done flags
    const size_t bytes = sizeof(SeqState) * batch;
    auto all_done = [&]() { for (int b = 0; b < batch; ++b) if (s->seq_host[b].active && !s->seq_host[b].done) return false; return true; };
    if (s->progress_cb) {
        // with a callback installed the host needs whole snapshots: no run-ahead, one synchronisation per 8 steps
        for (int step = 0; step < loop_count; step += kStepsPerGraph) {
            CHECK_CANCEL(s);
            hipGraphExec_t exec = nullptr;
            if (use_graphs()) { int r = get_step_graph(s, batch, step, &exec); if (r) return r; }
            if (exec) WH_HIP(hipGraphLaunch(exec, s->st));
            else for (int i = 0; i < kStepsPerGraph && step + i < loop_count; ++i) {
                DecodeBuffers db = whi::decode_buffers(s, batch, step + i);
                launch_decoder_step(db, s->cfg_dev, s->suppress_dev, true, s->st); WH_CHECK_LAUNCH();
            }
            WH_HIP(hipMemcpyAsync(s->seq_host, s->seq, bytes, hipMemcpyDeviceToHost, s->st));
            WH_HIP(hipStreamSynchronize(s->st));
            if (all_done()) break;
            int r = report_progress(s, batch);
            if (r) return r;
            if (all_done()) break;
        }
        WH_HIP(hipMemcpyAsync(s->seq_host, s->seq, bytes, hipMemcpyDeviceToHost, s->st));
        WH_HIP(hipStreamSynchronize(s->st));
        return WH_OK;
    }
    if (use_graphs()) {
        const int n_graphs = (loop_count + kStepsPerGraph - 1) / kStepsPerGraph;
        // WH_DBG_HOST=1: host-side cost of the replay loop (time inside hipGraphLaunch vs waiting for the device), one line per decode
        static const bool dbg_host = [] { const char* e = getenv("WH_DBG_HOST"); return e && e[0] == '1'; }();
        double t_launch = 0.0, t_wait = 0.0;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_begin = dbg_host ? now() : 0.0;
        struct Report { bool on; double *l, *w, t0; int n; decltype(now)* clk; ~Report() { if (on) fprintf(stderr, "[wh host] decode loop: %d graph launches, %.2f ms inside hipGraphLaunch, %.2f ms waiting for the device, %.2f ms total\n", n, *l * 1e3, *w * 1e3, ((*clk)() - t0) * 1e3); } } report{dbg_host, &t_launch, &t_wait, t_begin, n_graphs, &now};
        for (int g = 0; g < n_graphs; ++g) {
            if (cancelled(s)) { hipStreamSynchronize(s->st); return set_error(WH_ERR_CANCELLED, "decodeText: cancelled through the session's cancel flag"); }
            hipGraphExec_t exec;
            int r = get_step_graph(s, batch, g * kStepsPerGraph, &exec);
            if (r) return r;
            const double ta = dbg_host ? now() : 0.0;
            WH_HIP(hipGraphLaunch(exec, s->st));
            if (dbg_host) t_launch += now() - ta;
            // snapshot the slot states behind graph g; while it runs, look at the snapshot behind graph g-1
            // (at most one graph of run-ahead; `done` is monotonic, so a torn snapshot is harmless)
            WH_HIP(hipMemcpyAsync(s->seq_host, s->seq, bytes, hipMemcpyDeviceToHost, s->st));
            WH_HIP(hipEventRecord(s->ev[g & 1], s->st));
            if (g >= 1) {
                const double tb = dbg_host ? now() : 0.0;
                WH_HIP(hipEventSynchronize(s->ev[(g - 1) & 1]));
                if (dbg_host) t_wait += now() - tb;
                if (all_done()) break;
            }
        }
    } else {
        for (int step = 0; step < loop_count; ++step) {
            DecodeBuffers db = whi::decode_buffers(s, batch, step);
            launch_decoder_step(db, s->cfg_dev, s->suppress_dev, true, s->st);
            WH_CHECK_LAUNCH();
            if ((step & 7) == 7 && step + 1 < loop_count) {
                WH_HIP(hipMemcpyAsync(s->seq_host, s->seq, bytes, hipMemcpyDeviceToHost, s->st));
                WH_HIP(hipStreamSynchronize(s->st));
                if (all_done()) break;
            }
        }
    }
    WH_HIP(hipMemcpyAsync(s->seq_host, s->seq, bytes, hipMemcpyDeviceToHost, s->st));
    WH_HIP(hipStreamSynchronize(s->st));
    return WH_OK;
}

static int decode_text_impl(wh_session* s, int batch, const wh_decoding_options* opt, const wh_special_tokens* st, const int32_t* prompt,
                            int n_prompt, const int32_t* language_tokens, const float* temperatures, const int32_t* active, uint64_t seed,
                            wh_decoding_result* out);
extern "C" int wh_decode_text(wh_session* s, int batch, const wh_decoding_options* opt, const wh_special_tokens* st, const int32_t* prompt,
                              int n_prompt, const float* temperatures, const int32_t* active, uint64_t seed, wh_decoding_result* out) {
    return decode_text_impl(s, batch, opt, st, prompt, n_prompt, nullptr, temperatures, active, seed, out);
}
extern "C" int wh_decode_text_languages(wh_session* s, int batch, const wh_decoding_options* opt, const wh_special_tokens* st, const int32_t* prompt,
                                        int n_prompt, const int32_t* language_tokens, const float* temperatures, const int32_t* active, uint64_t seed,
                                        wh_decoding_result* out) {
    return decode_text_impl(s, batch, opt, st, prompt, n_prompt, language_tokens, temperatures, active, seed, out);
}
static int decode_text_impl(wh_session* s, int batch, const wh_decoding_options* opt, const wh_special_tokens* st, const int32_t* prompt,
                            int n_prompt, const int32_t* language_tokens, const float* temperatures, const int32_t* active, uint64_t seed,
                            wh_decoding_result* out) {
    CHECK_SESSION(s); CHECK_BATCH(s, batch);
    if (!opt || !st || !prompt || !out) return set_error(WH_ERR_DECODING_FAILED, "wh_decode_text: null argument");
    if (n_prompt < 1 || n_prompt >= kMaxTok) return set_error(WH_ERR_PREFILL_FAILED, "wh_decode_text: prompt length %d out of range [1,%d)", n_prompt, kMaxTok);
    const int V = s->m->dims.n_vocab;
    for (int i = 0; i < n_prompt; ++i)
        if (prompt[i] < 0 || prompt[i] >= V) return set_error(WH_ERR_PREFILL_FAILED, "wh_decode_text: prompt token %d out of vocabulary", prompt[i]);
    const int prefilled_index = 0;   // decoderInputs.cacheLength after reset (Core/Models.swift:313)
    int r = whi::upload_sampler_cfg(s, opt, st, prefilled_index, n_prompt, 0, seed);
    if (r) return r;
    if (opt->word_timestamps && s->m->n_align > 0) { r = whi::ensure_align(s); if (r) return r; }
    s->align_enabled = opt->word_timestamps && s->align;
    // per-slot language token (batched windows of different audios, each with its own detected language): it replaces the
    // token that follows <|startoftranscript|> in the shared prompt (prefillDecoderInputs, TextDecoder.swift:183-188)
    int lang_pos = -1;
    if (language_tokens && wh_is_model_multilingual(s->m))
        for (int i = 0; i + 1 < n_prompt; ++i) if (prompt[i] == st->start_of_transcript_token) { lang_pos = i + 1; break; }
    for (int b = 0; b < batch; ++b) {
        SeqState& q = s->seq_host[b];
        memset(&q, 0, sizeof(q));
        for (int i = 0; i < n_prompt; ++i) q.tokens[i] = prompt[i];
        if (lang_pos >= 0 && language_tokens[b] >= 0 && language_tokens[b] < V) q.tokens[lang_pos] = language_tokens[b];
        q.n_tokens = n_prompt; q.token_index = prefilled_index; q.next_token = q.tokens[0]; q.prompt_len = n_prompt;
        q.active = active ? (active[b] != 0) : 1;
        q.temperature = f16_round(temperatures ? temperatures[b] : opt->temperature);
    }
    // fused greedy path: every active slot samples at T = 0 (filters + softmax statistics in the logits epilogue)
    s->fused_greedy = true;
    for (int b = 0; b < batch; ++b) if (s->seq_host[b].active && s->seq_host[b].temperature != 0.0f) s->fused_greedy = false;
    if (const char* e = getenv("WH_NO_FUSED_SAMPLER")) if (e[0] == '1') s->fused_greedy = false;
    WH_HIP(hipMemcpyAsync(s->seq, s->seq_host, sizeof(SeqState) * batch, hipMemcpyHostToDevice, s->st));
    launch_rules_init(s->cfg_dev, s->seq, batch, s->st);
    const int loop_count = std::min(opt->sample_length, kMaxTok - 1);
    s->skip_special_in_progress = opt->skip_special_tokens != 0;
    s->special_begin_in_progress = st->special_token_begin;
    r = run_token_loop(s, batch, std::max(loop_count, 0));
    if (r) return r;
    for (int b = 0; b < batch; ++b) {
        if (!s->seq_host[b].active) { memset(&out[b], 0, sizeof(out[b])); continue; }
        whi::finalize_decoding_result(s->seq_host[b], opt, st, s->seq_host[b].temperature, &out[b]);
    }
    return WH_OK;
}

// decodeText with caller-supplied LogitsFiltering / TokenSampling objects: the reference's host loop (Core/TextDecoder.swift:573-757)
// over the step API - what the fused device loop cannot run (it knows the built-in filters and the greedy / top-k sampler only).
extern "C" int wh_decode_text_custom(wh_session* s, const wh_decoding_options* opt, const wh_special_tokens* st, const int32_t* prompt,
                                     int n_prompt, float temperature, uint64_t seed, const wh_logits_filter_fn* filters,
                                     void* const* filter_users, int n_filters, wh_token_sampler_fn sampler, void* sampler_user,
                                     wh_decoding_result* out) {
    CHECK_SESSION(s);
    if (!opt || !st || !prompt || !out) return set_error(WH_ERR_DECODING_FAILED, "wh_decode_text_custom: null argument");
    if (n_prompt < 1 || n_prompt >= kMaxTok) return set_error(WH_ERR_PREFILL_FAILED, "wh_decode_text_custom: prompt length %d out of range [1,%d)", n_prompt, kMaxTok);
    if (n_filters < 0 || (n_filters > 0 && !filters)) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_decode_text_custom: %d filters but no filter table", n_filters);
    const int V = s->m->dims.n_vocab;
    for (int i = 0; i < n_prompt; ++i)
        if (prompt[i] < 0 || prompt[i] >= V) return set_error(WH_ERR_PREFILL_FAILED, "wh_decode_text_custom: prompt token %d out of vocabulary", prompt[i]);
    try {
        int r = wh_reset_decoder_inputs(s, 1);
        if (r) return r;
        std::vector<float> logits((size_t)V);
        SeqState q;                                       // the loop state, in the layout finalize_decoding_result reads
        memset(&q, 0, sizeof(q));
        for (int i = 0; i < n_prompt; ++i) q.tokens[i] = prompt[i];
        q.n_tokens = n_prompt; q.prompt_len = n_prompt; q.active = 1;
        const float temp = f16_round(temperature);        // GreedyTokenSampler.temperature is FloatType (TokenSampler.swift:30)
        const int prefilled_index = 0, loop_count = std::min(opt->sample_length, kMaxTok - 1);
        const bool has_thr = !isnan(opt->first_token_log_prob_threshold);
        int next = prompt[n_prompt - 1];                  // :566 nextToken = currentTokens.last
        for (int ti = prefilled_index; ti < loop_count; ++ti) {
            CHECK_CANCEL(s);
            const bool is_prefill = ti < n_prompt - 1, is_last_prefill = ti == n_prompt - 1, is_first = ti == prefilled_index;
            if (ti < n_prompt) {                          // :581-594
                const bool is_ts = q.tokens[ti] >= st->time_token_begin, pred_ts = next >= st->time_token_begin;
                if (!(is_last_prefill && is_ts && pred_ts)) next = q.tokens[ti];
                else q.tokens[ti] = next;
            }
            const int32_t tok_in = next, pos_in = ti;
            r = wh_predict_logits(s, 1, &tok_in, &pos_in, logits.data());                       // :611-633
            if (r) return r;
            if (opt->float16_logits) for (auto& v : logits) v = (float)(_Float16)v;             // FloatType logits (Core/Models.swift:1041)
            q.steps += 1;
            for (int f = 0; f < n_filters; ++f)                                                 // custom filters come first (:860-862)
                if (filters[f]) filters[f](filter_users ? filter_users[f] : nullptr, logits.data(), V, q.tokens, q.n_tokens);
            r = wh_filter_logits(s, opt, st, q.tokens, q.n_tokens, prefilled_index, n_prompt, 0, logits.data(), V);   // :641-643
            if (r) return r;
            int32_t tok = 0; float lp = 0.0f; bool completed;
            if (sampler) {
                completed = sampler(sampler_user, logits.data(), V, q.tokens, q.logprobs, q.n_tokens, &tok, &lp) != 0;       // :652
                if (tok < 0 || tok >= V) return set_error(WH_ERR_DECODING_FAILED, "wh_decode_text_custom: the sampler returned token %d (vocabulary %d)", tok, V);
            } else {
                r = wh_sample_token(s, logits.data(), V, temp, opt->top_k, seed, ti, &tok, &lp);
                if (r) return r;
                completed = tok == st->end_token;
            }
            next = tok;
            const bool too_low = is_first && has_thr && lp < opt->first_token_log_prob_threshold;   // :662-667
            q.first_token_too_low = too_low ? 1 : 0;
            if (completed || q.n_tokens >= kMaxTok - 1 || too_low) break;                        // :669-674
            if (!is_prefill) { q.tokens[q.n_tokens] = tok; q.logprobs[q.n_tokens] = lp; q.n_tokens += 1; }   // :682-686
            if (s->progress_cb && !is_prefill) {                                                  // :723-755
                s->seq_host[0] = q;
                s->skip_special_in_progress = opt->skip_special_tokens != 0;
                s->special_begin_in_progress = st->special_token_begin;
                SeqState keep = q;
                r = report_progress(s, 1);
                if (r) return r;
                const bool stop = s->seq_host[0].done != 0;
                q = keep;
                if (stop) break;
            }
        }
        whi::finalize_decoding_result(q, opt, st, temp, out);
    } catch (const std::bad_alloc&) {
        return set_error(WH_ERR_OUT_OF_MEMORY, "wh_decode_text_custom: out of host memory");
    }
    return WH_OK;
}

extern "C" int wh_detect_language(wh_session* s, int batch, const wh_special_tokens* st, int32_t* lang_out, float* lp_out) {
    // TextDecoder.detectLanguage: one step on SOT at position 0, LanguageLogitsFilter, greedy sample (no KV/state update)
    CHECK_SESSION(s); CHECK_BATCH(s, batch);
    if (!st || !lang_out) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_detect_language: null argument");
    wh_decoding_options o;
    wh_decoding_options_default(&o);
    int r = whi::upload_sampler_cfg(s, &o, st, 0, 0, 1, 0);
    if (r) return r;
    for (int b = 0; b < batch; ++b) {
        SeqState& q = s->seq_host[b];
        memset(&q, 0, sizeof(q));
        q.tokens[0] = st->start_of_transcript_token; q.n_tokens = 1; q.next_token = st->start_of_transcript_token; q.active = 1;
    }
    WH_HIP(hipMemcpyAsync(s->seq, s->seq_host, sizeof(SeqState) * batch, hipMemcpyHostToDevice, s->st));
    bool keep = s->align_enabled;
    s->align_enabled = false;
    DecodeBuffers db = whi::decode_buffers(s, batch, 0);
    s->align_enabled = keep;
    launch_decoder_step(db, nullptr, nullptr, false, s->st);
    launch_filter_sample(s->cfg_dev, s->suppress_dev, s->seq, s->logits, batch, s->tok_out_dev, s->lp_out_dev, s->st);
    WH_CHECK_LAUNCH();
    WH_HIP(hipMemcpyAsync(lang_out, s->tok_out_dev, sizeof(int) * batch, hipMemcpyDeviceToHost, s->st));
    std::vector<float> lp(batch);
    WH_HIP(hipMemcpyAsync(lp.data(), s->lp_out_dev, sizeof(float) * batch, hipMemcpyDeviceToHost, s->st));
    WH_HIP(hipStreamSynchronize(s->st));
    if (lp_out) for (int b = 0; b < batch; ++b) lp_out[b] = lp[b];
    return WH_OK;
}

// ------------------------------------------------------------------------------------------------ segments
extern "C" int wh_find_seek_point_and_segments(const wh_decoding_result* res, const wh_decoding_options* opt, const wh_special_tokens* st,
                                               int all_segments_count, int current_seek, int segment_size, int32_t* new_seek,
                                               wh_segment* segs, int capacity) {
    // SegmentSeeker.findSeekPointAndSegments (Core/Text/SegmentSeeker.swift:41-189); token_offset indexes res->tokens
    if (!res || !opt || !st || !new_seek) return -2;
    const int timeToken = st->time_token_begin;
    const float spt = 0.02f;   // WhisperKit.secondsPerTimeToken
    int seek = current_seek;
    const float timeOffset = (float)seek / (float)WH_SAMPLE_RATE;
    if (!isnan(opt->no_speech_threshold)) {
        bool skip = res->no_speech_prob > opt->no_speech_threshold;
        if (!isnan(opt->log_prob_threshold) && res->avg_logprob > opt->log_prob_threshold) skip = false;
        if (skip) { *new_seek = seek + segment_size; return -1; }
    }
    const int n = res->n_tokens;
    std::vector<char> isTs(n);
    for (int i = 0; i < n; ++i) isTs[i] = res->tokens[i] >= timeToken;
    auto last3 = [&](bool a, bool b, bool c) { return n >= 3 && isTs[n - 3] == a && isTs[n - 2] == b && isTs[n - 1] == c; };
    const bool single = last3(false, true, false), none = last3(false, false, false);
    std::vector<int> slices;
    bool prev = false;
    for (int i = 0; i < n; ++i) { if (prev && isTs[i]) slices.push_back(i); prev = isTs[i]; }
    std::vector<wh_segment> out;
    auto mk = [&](int off, int cnt, float start, float end) {
        wh_segment g{};
        g.id = all_segments_count + (int)out.size(); g.seek = current_seek; g.start = start; g.end = end; g.token_offset = off; g.n_tokens = cnt;
        g.temperature = res->temperature; g.avg_logprob = res->avg_logprob; g.compression_ratio = res->compression_ratio; g.no_speech_prob = res->no_speech_prob;
        out.push_back(g);
    };
    if (!slices.empty()) {
        if (single) { int li = 0; for (int i = 0; i < n; ++i) if (isTs[i]) li = i; slices.push_back(li + 1); }
        else if (none) slices.push_back(n);
        int lastStart = 0;
        for (int e : slices) {
            int first = -1, last = -1;
            for (int i = lastStart; i < e; ++i) if (res->tokens[i] >= timeToken) { if (first < 0) first = res->tokens[i]; last = res->tokens[i]; }
            mk(lastStart, e - lastStart, timeOffset + (float)(first - timeToken) * spt, timeOffset + (float)(last - timeToken) * spt);
            lastStart = e;
        }
        if (!none) {
            int lt = res->tokens[lastStart - (single ? 1 : 0)] - timeToken;
            seek += (int)((float)lt * spt * (float)WH_SAMPLE_RATE);
        } else seek += segment_size;
    } else {
        float dur = (float)segment_size / (float)WH_SAMPLE_RATE;
        for (int i = 0; i < n; ++i) if (res->tokens[i] > timeToken) dur = (float)(res->tokens[i] - timeToken) * spt;
        mk(0, n, timeOffset, timeOffset + dur);
        seek += segment_size;
    }
    *new_seek = seek;
    if (segs) {
        if ((int)out.size() > capacity) return -2;
        for (size_t i = 0; i < out.size(); ++i) segs[i] = out[i];
    }
    return (int)out.size();
}

// ------------------------------------------------------------------------------------------------ TranscribeTask.run
// wh_transcription: text.h

struct AudioJob {
    const float* pcm; int n;
    int index = 0;                                  // position of the audio in the caller's batch (hook argument, item status)
    const wh_decoding_options* opt = nullptr;       // this audio's options (clip timestamps are per audio; everything else is shared by its group)
    int status = WH_OK; std::string error;          // Result<[TranscriptionResult], Error> of this audio (WhisperKit.swift:786-790)
    std::vector<std::pair<int, int>> clips;
    size_t clip = 0; int seek = 0; bool started = false; bool finished = false;
    int windows = 0;
    wh_transcription* tr;
    int cur_seek = 0, cur_size = 0;   // window in flight
};

static bool job_next_window(AudioJob& j, const wh_decoding_options* opt) {
    // advance to the next (clip, seek) that satisfies `seek < clipEnd - windowPadding` (TranscribeTask.swift:105-116)
    const int windowPadding = (int)(opt->window_clip_time * (float)WH_SAMPLE_RATE);
    while (j.clip < j.clips.size()) {
        if (!j.started) { j.seek = j.clips[j.clip].first; j.started = true; }
        if (j.seek < j.clips[j.clip].second - windowPadding) {
            j.cur_seek = j.seek;
            j.cur_size = std::min({kWindowSamples, j.n - j.seek, j.clips[j.clip].second - j.seek});
            return true;
        }
        ++j.clip; j.started = false;
    }
    j.finished = true;
    return false;
}

static int transcribe_jobs(wh_session* s, std::vector<AudioJob>& jobs, const wh_decoding_options* opt, const wh_special_tokens* st) {
    const wh_model* m = s->m;
    const bool multilingual = wh_is_model_multilingual(m) != 0;
    const int detect = opt->detect_language < 0 ? !opt->use_prefill_prompt : opt->detect_language;
    const double t_start = now_s();
    const double cf_start = cf_absolute_time();
    for (auto& j : jobs) { const wh_decoding_options* jo = j.opt ? j.opt : opt;      // clip timestamps belong to the audio, not to its group
                          j.clips = prepare_seek_clips(jo, j.n); j.tr->timings.input_audio_seconds = (double)j.n / WH_SAMPLE_RATE - (double)(jo->n_clip_timestamps > 0 && jo->clip_timestamps ? jo->clip_timestamps[0] : 0.0f);
                          j.tr->timings.pipeline_start = cf_start; }
    // temperature ladder in FloatType (TranscribeTask.swift:327)
    std::vector<float> temps;
    for (int i = 0; i <= opt->temperature_fallback_count; ++i)
        temps.push_back(f16_round(f16_round(opt->temperature) + f16_round(f16_round((float)i) * f16_round(opt->temperature_increment_on_fallback))));

    // beam search (no reference behaviour, wh_decode_text_beam): the T = 0 pass expands every window into beam_size slots
    const int beam = opt->beam_size > 1 ? opt->beam_size : 1;
    if (beam > 1 && opt->word_timestamps) return set_error(WH_ERR_INVALID_ARGUMENT, "beam_size > 1 cannot be combined with word_timestamps");
    if (beam > s->B) return set_error(WH_ERR_INVALID_ARGUMENT, "beam_size %d exceeds the session's %d slots", beam, s->B);
    const int round_cap = s->B / beam;
    std::vector<int> slot_job;   // job index per slot for the current round
    while (true) {
        // gather up to B jobs that still have a window (each job contributes one window per round: windows of one audio are sequential)
        // A window whose padOrTrim fails (TranscribeTask.swift:126-127 throws audioProcessingFailed) fails ITS audio only: the audio leaves the
        // batch with its status and message, its neighbours go on (the reference wraps every audio in its own Result, WhisperKit.swift:786-790).
        slot_job.clear();
        CHECK_CANCEL(s);
        double t0 = now_s();
        for (size_t ji = 0; ji < jobs.size() && (int)slot_job.size() < round_cap; ++ji) {
            AudioJob& j = jobs[ji];
            if (j.finished || j.status != WH_OK || !job_next_window(j, opt)) continue;
            const int b = (int)slot_job.size();
            // a window without samples: the reference slices audioArray[seek ..< seek + segmentSize], AudioProcessor.padOrTrimAudio returns nil for
            // the empty slice ("startIndex is outside the buffer size", Core/Audio/AudioProcessor.swift:151-155) and the task throws
            // transcriptionFailed("Audio samples are nil") (Core/TranscribeTask.swift:126-129) - a clip that ends beyond the samples gets here
            int r = j.cur_size > 0 ? wh_set_audio(s, b, j.pcm + j.cur_seek, j.cur_size)
                                   : set_error(WH_ERR_TRANSCRIPTION_FAILED, "audio %d: Audio samples are nil (window start %d is outside the buffer size %d)", j.index, j.cur_seek, j.n);
            if (r == WH_ERR_HIP) return r;                                 // the device is gone: every audio of the group fails
            if (r) { j.status = r; j.error = wh_last_error(); j.finished = true; continue; }
            slot_job.push_back((int)ji);
            if (s->hooks.window_preprocess)                                // TranscribeTask.windowPreprocess (:130)
                s->hooks.window_preprocess(s->hooks.user, j.index, j.pcm + j.cur_seek, j.cur_seek, j.cur_size);
            j.tr->seeks.push_back(j.cur_seek);
        }
        if (slot_job.empty()) break;
        const int nb = (int)slot_job.size();
        double t1 = now_s();
        int r = wh_log_mel_spectrogram(s, nb); if (r) return r;
        hipStreamSynchronize(s->st);
        double t2 = now_s();
        r = wh_encode_features(s, nb); if (r) return r;
        r = wh_prepare_decoder_inputs(s, nb); if (r) return r;
        hipStreamSynchronize(s->st);
        double t3 = now_s();
        // ---- decodeWithFallback (TranscribeTask.swift:316-411), all slots in lock step per temperature
        std::vector<wh_decoding_result> res(nb), tmp(nb);
        std::vector<int32_t> active(nb, 1);
        std::vector<int32_t> prompt(kMaxPrompt);
        int n_prompt = 1;
        prompt[0] = st->start_of_transcript_token;
        // language token per slot: every audio of the batch detects (and is prompted with) its own language, like the reference's
        // one TranscribeTask per audio (WhisperKit.swift:735-792); -1 = the options' language / English default of prefillDecoderInputs
        std::vector<int32_t> lang_slot(nb, opt->language_token);
        for (size_t ti = 0; ti < temps.size(); ++ti) {
            CHECK_CANCEL(s);
            std::vector<float> tv(nb, temps[ti]);
            bool per_slot_lang = false;
            if (multilingual && opt->language_token < 0 && detect) {
                std::vector<int32_t> lt(nb); std::vector<float> ll(nb);
                r = wh_detect_language(s, nb, st, lt.data(), ll.data()); if (r) return r;
                for (int b = 0; b < nb; ++b) {
                    if (!active[b]) continue;
                    lang_slot[b] = lt[b];
                    wh_transcription* t = jobs[slot_job[b]].tr;
                    if (!t->language_set) { t->language_token = lt[b]; t->language_set = true; }
                }
                per_slot_lang = true;
            }
            if (opt->use_prefill_prompt) {
                n_prompt = wh_prefill_prompt(m, opt, st, opt->language_token, prompt.data(), (int)prompt.size());
                if (n_prompt <= 0) return set_error(WH_ERR_PREFILL_FAILED, "prefill prompt does not fit");
            }
            r = whi::reset_decoder_inputs_masked(s, nb, active.data()); if (r) return r;     // accepted slots keep their alignment rows
            uint64_t seed = opt->seed + 1000003ull * (uint64_t)jobs[slot_job[0]].windows + ti;
            const int32_t* langs = (per_slot_lang && opt->use_prefill_prompt) ? lang_slot.data() : nullptr;
            bool all_active = true;
            for (int b = 0; b < nb; ++b) all_active &= active[b] != 0;
            if (beam > 1 && temps[ti] == 0.0f && all_active) {
                r = wh_decode_text_beam(s, nb, beam, opt->beam_patience, opt, st, prompt.data(), n_prompt, langs, tmp.data());
                if (r) return r;
                bool again = false;
                for (int b = 0; b < nb; ++b) again |= tmp[b].needs_fallback != 0;
                if (again && ti + 1 < temps.size()) { r = wh_prepare_decoder_inputs(s, nb); if (r) return r; }   // the beam slots overwrote the windows' cross K/V
            } else {
                r = decode_text_impl(s, nb, opt, st, prompt.data(), n_prompt, langs, tv.data(), active.data(), seed, tmp.data());
                if (r) return r;
            }
            bool any = false;
            for (int b = 0; b < nb; ++b) {
                if (!active[b]) continue;
                res[b] = tmp[b];
                AudioJob& j = jobs[slot_job[b]];
                j.tr->timings.total_decoding_loops += tmp[b].steps;
                if (tmp[b].needs_fallback && ti + 1 < temps.size()) { any = true; j.tr->timings.total_decoding_fallbacks += 1; }
                else active[b] = 0;
            }
            if (!any) break;
        }
        double t4 = now_s();
        // ---- windowing (TranscribeTask.swift:175-278)
        for (int b = 0; b < nb; ++b) {
            AudioJob& j = jobs[slot_job[b]];
            wh_transcription* tr = j.tr;
            // "Windowing" (TranscribeTask.swift:175-265) is host-only code shared with the CPU tests: wh_transcription_add_window
            std::vector<float> full;
            const float* alignment = nullptr;
            if (opt->word_timestamps && s->align) {
                full.resize((size_t)kMaxTok * kCtx);
                r = wh_get_alignment_weights(s, b, full.data()); if (r) return r;
                alignment = full.data();
            }
            const double windows_before = tr->timings.total_decoding_windows;
            int32_t seek = j.seek;
            const int seg_before = (int)tr->segments.size();
            r = wh_transcription_add_window(tr, s->tok, opt, st, &res[b], alignment, lang_slot[b], j.cur_size, &seek); if (r) return r;
            j.seek = seek;
            if (tr->timings.total_decoding_windows > windows_before) {      // the window had segments (`guard let currentSegments`, :239-242)
                int n_new = (int)tr->segments.size() - seg_before;
                if (s->hooks.window_postprocess) {                         // TranscribeTask.windowPostProcess (:246-250)
                    const int keep = s->hooks.window_postprocess(s->hooks.user, j.index, j.cur_seek, j.cur_size, tr, seg_before, n_new);
                    if (keep >= 0 && keep < n_new) { whi::transcription_truncate_segments(tr, seg_before + keep); n_new = keep; }
                }
                if (s->hooks.segment_discovery) s->hooks.segment_discovery(s->hooks.user, j.index, tr, seg_before, n_new);   // segmentDiscoveryCallback (:260)
            }
            if (tr->timings.total_decoding_windows > windows_before) j.windows += 1;
            tr->timings.audio_processing += (t1 - t0) / nb; tr->timings.logmels += (t2 - t1) / nb; tr->timings.encoding += (t3 - t2) / nb;
            tr->timings.decoding_loop += (t4 - t3) / nb; tr->timings.total_logmel_runs += 1; tr->timings.total_encoding_runs += 1;
            tr->timings.total_audio_processing_runs += 1;
            tr->timings.decoding_windowing += (now_s() - t4) / nb;
            if (tr->timings.first_token_time == 0) tr->timings.first_token_time = cf_start + (t4 - t_start);   // first decode of this audio done
        }
    }
    for (auto& j : jobs) {
        if (j.status != WH_OK) continue;
        wh_transcription* tr = j.tr;
        tr->timings.full_pipeline = now_s() - t_start;
        int fr = wh_transcription_finalize(tr, s->tok, opt, st);
        if (fr) { j.status = fr; j.error = wh_last_error(); }
    }
    return WH_OK;
}

// Two option sets may share a device batch when everything the lock-stepped loop reads is equal: all scalars and the prompt / prefix /
// suppress token lists (NaN == NaN: both nil).  Clip timestamps are positions inside ONE audio and are applied per audio.
static bool same_group_options(const wh_decoding_options& a, const wh_decoding_options& b) {
    auto fe = [](float x, float y) { return (isnan(x) && isnan(y)) || x == y; };
    auto le = [](const int32_t* x, int nx, const int32_t* y, int ny) {
        const int ex = x ? nx : 0, ey = y ? ny : 0;
        return (x == nullptr) == (y == nullptr) && ex == ey && (ex == 0 || memcmp(x, y, sizeof(int32_t) * (size_t)ex) == 0);
    };
    return a.task == b.task && a.language_token == b.language_token && fe(a.temperature, b.temperature) &&
           fe(a.temperature_increment_on_fallback, b.temperature_increment_on_fallback) && a.temperature_fallback_count == b.temperature_fallback_count &&
           a.sample_length == b.sample_length && a.top_k == b.top_k && a.use_prefill_prompt == b.use_prefill_prompt && a.detect_language == b.detect_language &&
           a.skip_special_tokens == b.skip_special_tokens && a.without_timestamps == b.without_timestamps && a.word_timestamps == b.word_timestamps &&
           fe(a.max_initial_timestamp, b.max_initial_timestamp) && a.max_window_seek == b.max_window_seek && fe(a.window_clip_time, b.window_clip_time) &&
           le(a.prompt_tokens, a.n_prompt_tokens, b.prompt_tokens, b.n_prompt_tokens) && le(a.prefix_tokens, a.n_prefix_tokens, b.prefix_tokens, b.n_prefix_tokens) &&
           a.suppress_blank == b.suppress_blank && le(a.suppress_tokens, a.n_suppress_tokens, b.suppress_tokens, b.n_suppress_tokens) &&
           fe(a.compression_ratio_threshold, b.compression_ratio_threshold) && fe(a.log_prob_threshold, b.log_prob_threshold) &&
           fe(a.first_token_log_prob_threshold, b.first_token_log_prob_threshold) && fe(a.no_speech_threshold, b.no_speech_threshold) && a.seed == b.seed &&
           a.float16_logits == b.float16_logits && a.beam_size == b.beam_size && fe(a.beam_patience, b.beam_patience);
}

// WhisperKit.transcribeWithOptions(audioArrays:decodeOptionsArray:) (Core/WhisperKit.swift:716-812): one DecodingOptions and one
// Result<[TranscriptionResult], Error> per audio.  Audios whose options can share a lock-stepped device batch (same_group_options) form
// a group; groups run one after the other (the reference runs one TranscribeTask per audio: grouping changes the batching, not a result).
// A failure that belongs to one audio (bad buffer, a window outside the samples) fails that audio; a failure of a group's shared state
// (invalid option combination, a device error) fails the group's audios; the other groups still run.
static int transcribe_items(wh_session* s, const float* const* pcm, const int32_t* n_samples, int n_audio, const wh_decoding_options* const* opts,
                            const wh_decoding_options* shared, const wh_special_tokens* st, wh_transcription** out, int32_t* statuses) {
    wh_decoding_options dflt;
    wh_decoding_options_default(&dflt);
    s->item_status.assign((size_t)n_audio, WH_OK);
    s->item_error.assign((size_t)n_audio, std::string());
    std::vector<AudioJob> all((size_t)n_audio);
    std::vector<int> group_of((size_t)n_audio, -1);
    std::vector<const wh_decoding_options*> group_opt;
    for (int i = 0; i < n_audio; ++i) {
        AudioJob& j = all[(size_t)i];
        j.index = i; j.pcm = pcm[i]; j.n = n_samples[i]; j.tr = nullptr;
        j.opt = (opts && opts[i]) ? opts[i] : (shared ? shared : &dflt);
        out[i] = nullptr;
        if (j.n < 0 || (j.n > 0 && !j.pcm)) {
            j.status = set_error(WH_ERR_AUDIO_PROCESSING_FAILED, "audio %d: invalid buffer", i); j.error = wh_last_error();
            continue;
        }
        int g = -1;
        for (size_t k = 0; k < group_opt.size() && g < 0; ++k) if (same_group_options(*group_opt[k], *j.opt)) g = (int)k;
        if (g < 0) { g = (int)group_opt.size(); group_opt.push_back(j.opt); }
        group_of[(size_t)i] = g;
    }
    int hard = WH_OK;
    for (size_t g = 0; g < group_opt.size(); ++g) {
        std::vector<AudioJob> jobs;
        for (int i = 0; i < n_audio; ++i) if (group_of[(size_t)i] == (int)g) { jobs.push_back(all[(size_t)i]); jobs.back().tr = new wh_transcription(); }
        const int r = hard ? hard : transcribe_jobs(s, jobs, group_opt[g], st);       // after a device error nothing else can run
        const std::string why = r ? wh_last_error() : "";
        if (r == WH_ERR_HIP || r == WH_ERR_CANCELLED) hard = r;
        for (auto& j : jobs) {
            AudioJob& a = all[(size_t)j.index];
            a.status = r ? r : j.status;
            a.error = r ? why : j.error;
            if (a.status == WH_OK) out[j.index] = j.tr; else delete j.tr;
        }
    }
    int n_ok = 0, first_bad = WH_OK;
    for (int i = 0; i < n_audio; ++i) {
        const AudioJob& a = all[(size_t)i];
        s->item_status[(size_t)i] = a.status; s->item_error[(size_t)i] = a.error;
        if (statuses) statuses[i] = a.status;
        if (a.status == WH_OK) ++n_ok; else if (first_bad == WH_OK) first_bad = a.status;
    }
    if (hard) return set_error(hard, "%s", s->item_error[0].empty() ? "wh_transcribe_batch: device failure" : s->item_error[0].c_str());
    if (n_ok == 0 && !statuses) {       // a caller that did not ask for per-audio statuses still learns why nothing came back
        for (int i = 0; i < n_audio; ++i) if (s->item_status[(size_t)i] == first_bad) return set_error(first_bad, "%s", s->item_error[(size_t)i].c_str());
    }
    return WH_OK;
}

extern "C" int wh_transcribe_batch_with_options(wh_session* s, const float* const* pcm, const int32_t* n_samples, int n_audio,
                                                const wh_decoding_options* const* opts, const wh_special_tokens* st, wh_transcription** out,
                                                int32_t* statuses) {
    CHECK_SESSION(s);
    if (!pcm || !n_samples || n_audio < 1 || !st || !out) return set_error(WH_ERR_TRANSCRIPTION_FAILED, "wh_transcribe_batch_with_options: null argument");
    WH_TRY
    return transcribe_items(s, pcm, n_samples, n_audio, opts, nullptr, st, out, statuses);
    WH_CATCH("wh_transcribe_batch_with_options")
}

// WhisperKit.transcribe(audioArrays:) -> [[TranscriptionResult]?] (Core/WhisperKit.swift:660-688 over :716-812): ONE options value for every
// audio; an audio that fails leaves out[i] == NULL (the reference's nil) and does not fail its neighbours; its status and message stay on
// the session (wh_session_item_status / wh_session_item_error).  The call itself fails only when nothing could run (every audio failed,
// the device is gone, cancellation).
extern "C" int wh_transcribe_batch(wh_session* s, const float* const* pcm, const int32_t* n_samples, int n_audio,
                                   const wh_decoding_options* opt, const wh_special_tokens* st, wh_transcription** out) {
    CHECK_SESSION(s);
    if (!pcm || !n_samples || n_audio < 1 || !opt || !st || !out) return set_error(WH_ERR_TRANSCRIPTION_FAILED, "wh_transcribe_batch: null argument");
    WH_TRY
    return transcribe_items(s, pcm, n_samples, n_audio, nullptr, opt, st, out, nullptr);
    WH_CATCH("wh_transcribe_batch")
}

extern "C" int wh_session_item_status(const wh_session* s, int audio_index) {
    if (!s || audio_index < 0 || (size_t)audio_index >= s->item_status.size()) return WH_ERR_INVALID_ARGUMENT;
    return s->item_status[(size_t)audio_index];
}
extern "C" const char* wh_session_item_error(const wh_session* s, int audio_index) {
    if (!s || audio_index < 0 || (size_t)audio_index >= s->item_error.size()) return "";
    return s->item_error[(size_t)audio_index].c_str();
}

extern "C" int wh_transcribe(wh_session* s, const float* pcm, int n, const wh_decoding_options* opt, const wh_special_tokens* st, wh_transcription** out) {
    const float* p[1] = {pcm};
    int32_t nn[1] = {n};
    return wh_transcribe_batch(s, p, nn, 1, opt, st, out);          // one audio: its failure is the call's failure
}

extern "C" int wh_transcribe_chunked(wh_session* s, const float* pcm, int n, const wh_decoding_options* opt, const wh_special_tokens* st,
                                     wh_transcription** out, int capacity, int32_t* seek_offsets_out, int* n_out) {
    CHECK_SESSION(s);
    if (!opt || !st || !out || !n_out || capacity < 1 || n < 0 || (n > 0 && !pcm))
        return set_error(WH_ERR_TRANSCRIPTION_FAILED, "wh_transcribe_chunked: invalid argument");
    *n_out = 0;
    if (n <= kWindowSamples) {      // not chunkable: runTranscribeTask on the whole array (WhisperKit.swift:913-919)
        int r = wh_transcribe(s, pcm, n, opt, st, &out[0]);
        if (r) return r;
        if (seek_offsets_out) seek_offsets_out[0] = 0;
        *n_out = 1;
        return WH_OK;
    }
    int nc = wh_vad_chunk_all(pcm, n, kWindowSamples, opt, nullptr, nullptr, 0);
    if (nc < 0) return set_error(WH_ERR_AUDIO_PROCESSING_FAILED, "wh_transcribe_chunked: startIndex is outside the buffer size");
    if (nc > capacity) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_transcribe_chunked: %d chunks exceed the capacity %d", nc, capacity);
    std::vector<int32_t> cs(nc), ce(nc);
    wh_vad_chunk_all(pcm, n, kWindowSamples, opt, cs.data(), ce.data(), nc);
    wh_decoding_options chunked = *opt;     // "Reset the seek times since we've already chunked the audio" (:889-891)
    chunked.clip_timestamps = nullptr;
    chunked.n_clip_timestamps = 0;
    std::vector<const float*> ptrs(nc);
    std::vector<int32_t> lens(nc);
    for (int i = 0; i < nc; ++i) { ptrs[i] = pcm + cs[i]; lens[i] = ce[i] - cs[i]; }
    int r = wh_transcribe_batch(s, ptrs.data(), lens.data(), nc, &chunked, st, out);
    if (r) return r;
    int kept = 0;                                             // a chunk that failed is logged and skipped (AudioChunker.swift:19-37: `case .failure`)
    for (int i = 0; i < nc; ++i) {
        if (!out[i]) continue;
        wh_transcription_apply_seek_offset(out[i], cs[i]);   // updateSeekOffsetsForResults
        wh_transcription* t = out[i];
        out[i] = nullptr; out[kept] = t;
        if (seek_offsets_out) seek_offsets_out[kept] = cs[i];
        ++kept;
    }
    *n_out = kept;
    return WH_OK;
}

extern "C" int wh_session_set_progress_callback(wh_session* s, wh_progress_fn fn, void* user) {
    if (!s) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_session_set_progress_callback: null session");
    s->progress_cb = fn;
    s->progress_user = user;
    return WH_OK;
}

extern "C" int wh_session_set_window_hooks(wh_session* s, const wh_window_hooks* hooks) {
    if (!s) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_session_set_window_hooks: null session");
    s->hooks = hooks ? *hooks : wh_window_hooks{};
    return WH_OK;
}

extern "C" int wh_session_set_tokenizer(wh_session* s, const wh_tokenizer* t) {
    if (!s) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_session_set_tokenizer: null session");
    s->tok = t;
    return WH_OK;
}

extern "C" void wh_transcription_free(wh_transcription* t) { delete t; }
extern "C" int wh_transcription_n_segments(const wh_transcription* t) { return t ? (int)t->segments.size() : -1; }
extern "C" int wh_transcription_segment(const wh_transcription* t, int i, wh_segment* out) {
    if (!t || !out || i < 0 || i >= (int)t->segments.size()) return set_error(WH_ERR_INVALID_ARGUMENT, "segment index out of range");
    *out = t->segments[i];
    return WH_OK;
}
extern "C" int wh_transcription_n_words(const wh_transcription* t) { return t ? (int)t->words.size() : -1; }
extern "C" int wh_transcription_word(const wh_transcription* t, int i, wh_word_timing* out) {
    if (!t || !out || i < 0 || i >= (int)t->words.size()) return set_error(WH_ERR_INVALID_ARGUMENT, "word index out of range");
    *out = t->words[i];
    return WH_OK;
}
extern "C" int wh_transcription_tokens(const wh_transcription* t, const int32_t** tokens, const float** logprobs, int* n) {
    if (!t || !n) return set_error(WH_ERR_INVALID_ARGUMENT, "null transcription");
    if (tokens) *tokens = t->tokens.data();
    if (logprobs) *logprobs = t->logprobs.data();
    *n = (int)t->tokens.size();
    return WH_OK;
}
extern "C" int wh_transcription_language_token(const wh_transcription* t) { return t ? t->language_token : -1; }
extern "C" int wh_transcription_timings(const wh_transcription* t, wh_timings* out) {
    if (!t || !out) return set_error(WH_ERR_INVALID_ARGUMENT, "null transcription");
    *out = t->timings;
    return WH_OK;
}
extern "C" int wh_transcription_window_seeks(const wh_transcription* t, const int32_t** seeks, int* n) {
    if (!t || !n) return set_error(WH_ERR_INVALID_ARGUMENT, "null transcription");
    if (seeks) *seeks = t->seeks.data();
    *n = (int)t->seeks.size();
    return WH_OK;
}

// ------------------------------------------------------------------------------------------------ measurement hook
static const char* kKindNames[KK_COUNT] = {
    "mel_power", "mel_finalize", "gemm_conv1", "gemm_conv2", "layernorm", "gemm_enc_qkv", "enc_attention", "gemm_enc_o", "gemm_enc_fc1",
    "gemm_enc_fc2", "gemm_cross_kv", "dec_proj_qkv", "dec_self_attn", "dec_proj_oproj", "dec_proj_cq", "dec_cross_attn", "dec_proj_coproj",
    "dec_proj_fc1", "dec_proj_fc2", "dec_proj_logits", "sampler", "dec_embed", "dec_xabs_qk", "dec_xabs_vup"};
namespace wh { unsigned long long* debug_buffer(); }
extern "C" int wh_debug_dump(const char* path) {
    unsigned long long* b = wh::debug_buffer();
    if (!b || !path) return -1;
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)KK_COUNT * 4096 * 8);
    if (hipMemcpy(h.data(), b, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    FILE* f = fopen(path, "wb");
    if (!f) return -3;
    fwrite(h.data(), 8, h.size(), f);
    fclose(f);
    return 0;
}
extern "C" int wh_kernel_kind_count(void) { return KK_COUNT; }
extern "C" const char* wh_kernel_kind_name(int kind) { return (kind >= 0 && kind < KK_COUNT) ? kKindNames[kind] : nullptr; }

extern "C" int wh_measure_kernels(wh_session* s, int batch, int n_steps, double* avg_us, int32_t* launches) {
    // One eager pass of the hot path on the session stream - log-mel, encoder, cross-K/V projection, then `n_steps` decoder
    // steps (state as left by the last wh_decode_text setup, re-armed at position 0) - with a HIP event pair around every
    // kernel launch; reports the average duration and the launch count per KernelKind (wh_kernel_kind_name).
    CHECK_SESSION(s); CHECK_BATCH(s, batch);
    if (!avg_us || !launches || n_steps < 0 || n_steps > kMaxTok - 2) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_measure_kernels: invalid argument");
    const int L = s->m->dims.n_text_layer, Le = s->m->dims.n_audio_layer;
    // launches per decoder step: embed + L x (8, or 10 in absorbed mode: xabs_qk / xabs_attn / xabs_vup replace dec_cross_attn) + logits + sampler
    const size_t cap = (size_t)((s->use_xabs ? 10 : 8) * L + 4) * n_steps + 7 * Le + 16;
    KernelProfiler prof;
    prof.ev.resize(2 * cap); prof.kind.resize(cap); prof.capacity = cap;
    for (auto& e : prof.ev) WH_HIP(hipEventCreate(&e));
    WH_HIP(hipMemcpyAsync(s->seq_host, s->seq, sizeof(SeqState) * batch, hipMemcpyDeviceToHost, s->st));
    WH_HIP(hipStreamSynchronize(s->st));
    std::vector<SeqState> saved(s->seq_host, s->seq_host + batch);
    g_prof = &prof;
    int r = wh_log_mel_spectrogram(s, batch);
    if (!r) r = wh_encode_features(s, batch);
    if (!r) r = wh_prepare_decoder_inputs(s, batch);
    if (!r && n_steps > 0) {
        // re-arm the slots: keep prompt/config, restart the loop at position 0
        for (int b = 0; b < batch; ++b) {
            SeqState& q = s->seq_host[b];
            q = saved[b];
            int pl = std::max(q.prompt_len, 1);
            q.n_tokens = pl; q.token_index = 0; q.next_token = q.tokens[0]; q.done = 0; q.active = 1; q.steps = 0; q.first_token_too_low = 0;
        }
        hipMemcpyAsync(s->seq, s->seq_host, sizeof(SeqState) * batch, hipMemcpyHostToDevice, s->st);
        launch_rules_init(s->cfg_dev, s->seq, batch, s->st);
        for (int i = 0; i < n_steps; ++i) {
            DecodeBuffers db = whi::decode_buffers(s, batch, i);
            launch_decoder_step(db, s->cfg_dev, s->suppress_dev, true, s->st);
        }
    }
    g_prof = nullptr;
    hipError_t le = hipGetLastError();
    hipError_t se = hipStreamSynchronize(s->st);
    double tot[KK_COUNT] = {0};
    int cnt[KK_COUNT] = {0};
    for (size_t i = 0; i < prof.n; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, prof.ev[2 * i], prof.ev[2 * i + 1]) == hipSuccess) { tot[prof.kind[i]] += ms * 1000.0; cnt[prof.kind[i]]++; }
    }
    for (int k = 0; k < KK_COUNT; ++k) { avg_us[k] = cnt[k] ? tot[k] / cnt[k] : 0.0; launches[k] = cnt[k]; }
    for (auto& e : prof.ev) hipEventDestroy(e);
    if (r) return r;
    if (le != hipSuccess || se != hipSuccess) return set_error(WH_ERR_HIP, "wh_measure_kernels: %s", hipGetErrorString(le != hipSuccess ? le : se));
    return WH_OK;
}
