// Internal model / session structures behind the opaque C-ABI handles.
#pragma once
#include <map>
#include <new>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "kernels.h"
#include "text.h"
#include "whisperhip.h"

struct WhTensor {
    void* dev = nullptr;
    int dtype = 0, ndim = 0;
    long long shape[4] = {1, 1, 1, 1};
    size_t nbytes = 0;
};

struct EncLayerW {
    const float *ln1_g, *ln1_b; const f16* qkv_w; const float* qkv_b;
    const f16* o_w; const float* o_b;
    const float *ln2_g, *ln2_b; const f16* fc1_w; const float* fc1_b; const f16* fc2_w; const float* fc2_b;
};

struct wh_model {
    wh_dims dims{};
    int device = 0;
    void* blob_dev = nullptr;
    size_t blob_bytes = 0;
    std::unordered_map<std::string, WhTensor> t;
    wh::MelTables mel{};
    void* mel_tables_dev = nullptr;
    std::vector<EncLayerW> enc;
    std::vector<wh::DecLayerW> dec;
    const f16 *conv1_w, *conv2_w, *emb, *ckv_w;
    const float *conv1_b, *conv2_b, *enc_pos, *lnp_g, *lnp_b, *dec_pos, *ckv_b, *lnf_g, *lnf_b;
    // MFMA decode path (decoder32.hip): decoder weights re-tiled into MFMA A-fragment order + LayerNorm fold vectors, built at load
    std::vector<wh::Dec32LayerW> dec32;
    void* dec32_blob = nullptr;
    const f16* emb_t = nullptr; const float *lg_g = nullptr, *lg_c = nullptr;
    // weight-absorbed cross-attention (xabs.hip): W_k^T tiles + W_v tiles per layer, built by the first session that uses the path
    std::vector<wh::XabsLayerW> xabs;
    void* xabs_blob = nullptr;
    std::mutex xabs_mu;
    std::vector<int> align_slot;   // [L*H] -> slot or -1
    int n_align = 0;
    int* align_slot_dev = nullptr;
    // cross-attention gate (dec_shared.h): one device word per model; used by the step launches of a session while the model carries
    // more than one session (WH_XATT_GATE=0 never, =1 always)
    int* xattn_gate = nullptr;
    std::atomic<int> n_sessions{0};
};

// step graphs are keyed by everything their captured launches bake in
struct WhGraphKey {
    int batch, align, fused, n_align, self_rows, gate;
    bool operator<(const WhGraphKey& o) const {
        return std::tie(batch, align, fused, n_align, self_rows, gate) < std::tie(o.batch, o.align, o.fused, o.n_align, o.self_rows, o.gate);
    }
};

struct wh_session {
    wh_model* m = nullptr;
    int B = 0;
    hipStream_t st = nullptr;
    // stage buffers
    float* pcm = nullptr; int* n_valid = nullptr;
    float* logspec = nullptr; unsigned* maxkey = nullptr; f16* mel_t = nullptr; float* mel_f32 = nullptr;
    f16* h1 = nullptr; float* x = nullptr; f16* xn = nullptr; f16 *q16 = nullptr, *k16 = nullptr, *vt16 = nullptr, *att16 = nullptr;
    f16* hmlp = nullptr; f16* enc16 = nullptr; float* enc32 = nullptr;
    // decoder
    f16 *cross_k_hi = nullptr, *cross_v_hi = nullptr;               // K / V-row mode: 24-bit rows (kernels.h hr24), Float16 part ...
    signed char *cross_k_lo = nullptr, *cross_v_lo = nullptr;       // ... and 8-bit residuals
    f16 *self_k = nullptr, *self_v = nullptr;
    float *part = nullptr, *logits = nullptr;
    int* ticket = nullptr;
    float *align = nullptr, *align_mean = nullptr;
    int n_align_alloc = 0;                // alignment heads the `align` allocation was sized for
    int align_znorm = 0, align_median = 0;   // optional openai/whisper-style post-processing (wh_session_set_alignment_postprocess)
    float* align_tmp = nullptr; int align_tmp_heads = 0;   // [224][n_align][1500] softmax rows + [2][n_align][1500] statistics + 224 flags
    std::map<WhGraphKey, hipGraphExec_t> graphs;   // captured 8-step decode graphs of THIS session (no process-wide state)
    std::map<WhGraphKey, unsigned long long> graph_use;   // last use (a counter) per graph: the cache is capped, least recently used configuration first
    unsigned long long graph_tick = 0;
    const volatile int32_t* cancel_flag = nullptr; // polled between step graphs and pipeline stages (Task.checkCancellation)
    bool use_xabs = false;                // cross-attention path of this session (fixed at creation: never a function of the live batch)
    wh::Xabs xabs{};                      // absorbed queries + split partials (one allocation: xabs_blob)
    void* xabs_blob = nullptr;
    wh::Dec32 d32{};                      // decode-step activations: residual, planes, split-K scratch (one allocation: d32_blob)
    void* d32_blob = nullptr;
    wh::SeqState* seq = nullptr;
    wh::SeqState* seq_host = nullptr;     // pinned
    wh::SamplerCfg* cfg_dev = nullptr;
    int* suppress_dev = nullptr;
    unsigned char* sup_mask_dev = nullptr;   // [V] SuppressTokensFilter byte mask (fused greedy sampler)
    float* stats = nullptr;                  // [B][kStatBlocks][8]
    bool fused_greedy = false;
    int *tok_out_dev = nullptr; float* lp_out_dev = nullptr;
    float* scratch_logits = nullptr;       // [V] for the filter / sample KAT entry points
    // beam search (wh_decode_text_beam, allocated on first use): row -> owning slot table of the self-attention cache, top-k outputs
    int *beam_owner = nullptr, *beam_tok = nullptr; float* beam_lp = nullptr;
    hipEvent_t ev[8]{};
    bool align_enabled = false;
    wh_timings last_timings{};
    const wh_tokenizer* tok = nullptr;       // TextDecoding.tokenizer; not owned
    wh_progress_fn progress_cb = nullptr;    // TranscriptionCallback
    void* progress_user = nullptr;
    wh_window_hooks hooks{};                 // TranscribeTask.windowPreprocess / windowPostProcess / segmentDiscoveryCallback
    bool skip_special_in_progress = false;
    int special_begin_in_progress = 1 << 30;
    // per-audio Result of the last wh_transcribe_batch* call (WhisperKit.transcribeWithOptions returns one Result per audio, WhisperKit.swift:786-790)
    std::vector<int> item_status;
    std::vector<std::string> item_error;
};

namespace whi {
int set_error(int code, const char* fmt, ...);
#define WH_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return whi::set_error(WH_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define WH_CHECK_LAUNCH() WH_HIP(hipGetLastError())
// the HIP current device is per host thread: sessions are driven from worker threads, so every session entry point re-selects it
#define CHECK_SESSION(s) do { if (!(s) || !(s)->m) return whi::set_error(WH_ERR_MODELS_UNAVAILABLE, "%s: session/model is null (modelsUnavailable)", __func__); \
                              if (hipSetDevice((s)->m->device) != hipSuccess) return whi::set_error(WH_ERR_HIP, "%s: hipSetDevice(%d) failed", __func__, (s)->m->device); } while (0)
#define CHECK_SLOT(s, b) do { if ((b) < 0 || (b) >= (s)->B) return whi::set_error(WH_ERR_INVALID_ARGUMENT, "%s: slot %d out of range [0,%d)", __func__, (b), (s)->B); } while (0)
#define CHECK_BATCH(s, n) do { if ((n) < 1 || (n) > (s)->B) return whi::set_error(WH_ERR_INVALID_ARGUMENT, "%s: batch %d out of range [1,%d]", __func__, (n), (s)->B); } while (0)
// C++ exceptions never cross the C ABI: entry points that allocate on the host wrap their body in WH_TRY / WH_CATCH
#define WH_TRY try {
#define WH_CATCH(name) } catch (const std::bad_alloc&) { return whi::set_error(WH_ERR_OUT_OF_MEMORY, "%s: out of host memory", name); } \
                         catch (const std::exception& e_) { return whi::set_error(WH_ERR_INVALID_ARGUMENT, "%s: %s", name, e_.what()); }

wh::DecodeBuffers decode_buffers(wh_session* s, int batch, int max_position = wh::kMaxTok - 1);
void drop_session_graphs(wh_session* s);
int ensure_align(wh_session* s);          // (re)allocate the raw alignment-head score buffer for the model's current head set
int reset_decoder_inputs_masked(wh_session* s, int batch, const int32_t* active);
// host logic shared by wh_decode_text / wh_transcribe (host.hip)
void transcription_truncate_segments(wh_transcription* t, int n_keep);     // results.cpp: drop segments [n_keep, end) with their tokens / words
void finalize_decoding_result(const wh::SeqState& sq, const wh_decoding_options* opt, const wh_special_tokens* st,
                              float temperature, wh_decoding_result* out);
}  // namespace whi
