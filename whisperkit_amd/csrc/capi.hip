// C-ABI implementation: model / session objects and the three model-stage entry points
// (FeatureExtracting, AudioEncoding, TextDecoding of argmaxinc/WhisperKit) over the gfx950 kernels.
// See include/whisperhip.h for the reference interface each function replaces.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cstdlib>

#include "internal.h"

using namespace wh;

namespace wh { thread_local KernelProfiler* g_prof = nullptr; unsigned long long* debug_buffer(); }

namespace whi {
static thread_local char g_err[512] = "";
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace whi
using whi::set_error;

extern "C" const char* wh_last_error(void) { return whi::g_err; }
extern "C" const char* wh_version(void) { return "whisperhip 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------------ mel tables
static double hz_to_mel(double f) {   // slaney scale
    const double min_log_hz = 1000.0, min_log_mel = 15.0, logstep = 27.0 / log(6.4);
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) * logstep : 3.0 * f / 200.0;
}
static double mel_to_hz(double m) {
    const double min_log_hz = 1000.0, min_log_mel = 15.0, logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : 200.0 * m / 3.0;
}

static int build_mel_tables(wh_model* m) {
    const int n_mels = m->dims.n_mels;
    const size_t nb = (size_t)200 * kBinsPad, nf = (size_t)kBins * n_mels;
    std::vector<float> bc(nb, 0.f), bs(nb, 0.f), filt(nf, 0.f);
    std::vector<int2> rng(n_mels);
    for (int n = 1; n <= 200; ++n) {
        double w = 0.5 - 0.5 * cos(2.0 * M_PI * n / 400.0);
        for (int k = 0; k < kBins; ++k) {
            int ph = (int)(((long long)n * k) % 400);   // exact phase reduction
            double ang = 2.0 * M_PI * ph / 400.0;
            bc[(size_t)(n - 1) * kBinsPad + k] = (float)(w * cos(ang));
            bs[(size_t)(n - 1) * kBinsPad + k] = (n == 200) ? 0.0f : (float)(w * sin(ang));
        }
    }
    // slaney mel filterbank (transformers audio_utils.mel_filter_bank, norm="slaney", mel_scale="slaney")
    std::vector<double> ff(n_mels + 2);
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(8000.0);
    for (int i = 0; i < n_mels + 2; ++i) ff[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    for (int j = 0; j < n_mels; ++j) {
        int lo = kBins, hi = -1;
        double enorm = 2.0 / (ff[j + 2] - ff[j]);
        for (int k = 0; k < kBins; ++k) {
            double f = 8000.0 * k / (kBins - 1);
            double down = (f - ff[j]) / (ff[j + 1] - ff[j]);
            double up = (ff[j + 2] - f) / (ff[j + 2] - ff[j + 1]);
            double v = std::max(0.0, std::min(down, up)) * enorm;
            filt[(size_t)k * n_mels + j] = (float)v;
            if (v > 0) { lo = std::min(lo, k); hi = std::max(hi, k); }
        }
        if (hi < lo) { lo = 0; hi = -1; }
        rng[j] = int2{lo, hi};
    }
    // compact non-zero weights per filter (<= 2 filters overlap a bin, so <= ~2 * 201 values): the kernel keeps them in LDS
    std::vector<float> fc;
    std::vector<int> foff(n_mels);
    for (int j = 0; j < n_mels; ++j) {
        foff[j] = (int)fc.size();
        for (int k = rng[j].x; k <= rng[j].y; ++k) fc.push_back(filt[(size_t)k * n_mels + j]);
    }
    if (fc.size() > 1024) return set_error(WH_ERR_MODELS_UNAVAILABLE, "mel filterbank has %zu non-zeros (> 1024)", fc.size());
    const size_t o_filt = 2 * nb * 4, o_rng = o_filt + nf * 4, o_fc = o_rng + n_mels * sizeof(int2), o_foff = o_fc + fc.size() * 4;
    size_t bytes = o_foff + n_mels * sizeof(int);
    char* dev = nullptr;
    WH_HIP(hipMalloc((void**)&dev, bytes));
    m->mel_tables_dev = dev;
    WH_HIP(hipMemcpy(dev, bc.data(), nb * 4, hipMemcpyHostToDevice));
    WH_HIP(hipMemcpy(dev + nb * 4, bs.data(), nb * 4, hipMemcpyHostToDevice));
    WH_HIP(hipMemcpy(dev + o_filt, filt.data(), nf * 4, hipMemcpyHostToDevice));
    WH_HIP(hipMemcpy(dev + o_rng, rng.data(), n_mels * sizeof(int2), hipMemcpyHostToDevice));
    WH_HIP(hipMemcpy(dev + o_fc, fc.data(), fc.size() * 4, hipMemcpyHostToDevice));
    WH_HIP(hipMemcpy(dev + o_foff, foff.data(), n_mels * sizeof(int), hipMemcpyHostToDevice));
    m->mel.basis_c = (const float*)dev;
    m->mel.basis_s = (const float*)(dev + nb * 4);
    m->mel.filt = (const float*)(dev + o_filt);
    m->mel.filt_range = (const int2*)(dev + o_rng);
    m->mel.filt_c = (const float*)(dev + o_fc);
    m->mel.filt_off = (const int*)(dev + o_foff);
    m->mel.filt_nnz = (int)fc.size();
    m->mel.n_mels = n_mels;
    return WH_OK;
}

// ------------------------------------------------------------------------------------------------ model
#pragma pack(push, 1)
struct BlobEntry {
    char name[64];
    int32_t dtype, ndim;
    int64_t shape[4];
    int64_t offset, nbytes;
};
#pragma pack(pop)

// `count` = elements the kernels will read from this tensor (dims-derived): a blob whose table disagrees is rejected here
// instead of driving out-of-bounds device reads later.
template <typename T>
static int get_tensor(wh_model* m, const std::string& name, const T** out, int dtype, size_t count) {
    auto it = m->t.find(name);
    if (it == m->t.end()) return set_error(WH_ERR_MODELS_UNAVAILABLE, "weight blob has no tensor '%s'", name.c_str());
    if (it->second.dtype != dtype) return set_error(WH_ERR_MODELS_UNAVAILABLE, "tensor '%s' has dtype %d, expected %d", name.c_str(), it->second.dtype, dtype);
    if (it->second.nbytes != count * sizeof(T))
        return set_error(WH_ERR_MODELS_UNAVAILABLE, "tensor '%s' has %zu bytes, the model dimensions need %zu", name.c_str(), it->second.nbytes, count * sizeof(T));
    *out = (const T*)it->second.dev;
    return WH_OK;
}
#define GET16(name, field, count) do { int _r = get_tensor<f16>(m, name, &(field), 0, (size_t)(count)); if (_r) return _r; } while (0)
#define GET32(name, field, count) do { int _r = get_tensor<float>(m, name, &(field), 1, (size_t)(count)); if (_r) return _r; } while (0)

static int bind_weights(wh_model* m) {
    const wh_dims& D = m->dims;
    const size_t d = D.n_audio_state, nm = D.n_mels, V = D.n_vocab, L = D.n_text_layer;
    GET16("enc.conv1.w", m->conv1_w, d * 3 * nm); GET32("enc.conv1.b", m->conv1_b, d);
    GET16("enc.conv2.w", m->conv2_w, d * 3 * d); GET32("enc.conv2.b", m->conv2_b, d);
    GET32("enc.pos", m->enc_pos, (size_t)kCtx * d); GET32("enc.lnp.g", m->lnp_g, d); GET32("enc.lnp.b", m->lnp_b, d);
    m->enc.resize(D.n_audio_layer);
    for (int i = 0; i < D.n_audio_layer; ++i) {
        std::string p = "enc." + std::to_string(i);
        EncLayerW& w = m->enc[i];
        GET32(p + ".ln1.g", w.ln1_g, d); GET32(p + ".ln1.b", w.ln1_b, d); GET16(p + ".qkv.w", w.qkv_w, 3 * d * d); GET32(p + ".qkv.b", w.qkv_b, 3 * d);
        GET16(p + ".o.w", w.o_w, d * d); GET32(p + ".o.b", w.o_b, d); GET32(p + ".ln2.g", w.ln2_g, d); GET32(p + ".ln2.b", w.ln2_b, d);
        GET16(p + ".fc1.w", w.fc1_w, 4 * d * d); GET32(p + ".fc1.b", w.fc1_b, 4 * d); GET16(p + ".fc2.w", w.fc2_w, 4 * d * d); GET32(p + ".fc2.b", w.fc2_b, d);
    }
    GET16("dec.emb", m->emb, V * d); GET32("dec.pos", m->dec_pos, (size_t)D.n_text_ctx * d); GET16("dec.ckv.w", m->ckv_w, L * 2 * d * d); GET32("dec.ckv.b", m->ckv_b, L * 2 * d);
    GET32("dec.ln.g", m->lnf_g, d); GET32("dec.ln.b", m->lnf_b, d);
    m->dec.resize(D.n_text_layer);
    for (int i = 0; i < D.n_text_layer; ++i) {
        std::string p = "dec." + std::to_string(i);
        DecLayerW& w = m->dec[i];
        GET32(p + ".ln1.g", w.ln1_g, d); GET32(p + ".ln1.b", w.ln1_b, d); GET16(p + ".qkv.w", w.qkv_w, 3 * d * d); GET32(p + ".qkv.b", w.qkv_b, 3 * d);
        GET16(p + ".o.w", w.o_w, d * d); GET32(p + ".o.b", w.o_b, d);
        GET32(p + ".ln2.g", w.ln2_g, d); GET32(p + ".ln2.b", w.ln2_b, d); GET16(p + ".cq.w", w.cq_w, d * d); GET32(p + ".cq.b", w.cq_b, d);
        GET16(p + ".co.w", w.co_w, d * d); GET32(p + ".co.b", w.co_b, d);
        GET32(p + ".ln3.g", w.ln3_g, d); GET32(p + ".ln3.b", w.ln3_b, d);
        GET16(p + ".fc1.w", w.fc1_w, 4 * d * d); GET32(p + ".fc1.b", w.fc1_b, 4 * d); GET16(p + ".fc2.w", w.fc2_w, 4 * d * d); GET32(p + ".fc2.b", w.fc2_b, d);
    }
    return WH_OK;
}

// Carve 256-byte aligned sub-buffers out of one allocation.
struct Carver {
    char* base = nullptr; size_t off = 0;
    template <typename T> T* take(size_t n) { T* p = reinterpret_cast<T*>(base + off); off = (off + n * sizeof(T) + 255) / 256 * 256; return p; }
};

static int build_dec32(wh_model* m) {
    const wh_dims& D = m->dims;
    const size_t d = D.n_text_state, L = D.n_text_layer, V = D.n_vocab, Vp = (V + 31) / 32 * 32;
    size_t bytes = L * (14 * d * d * 2 + 16 * d * 4 + 12 * 256) + Vp * d * 2 + 2 * Vp * 4 + 3 * 256;
    hipError_t e = hipMalloc(&m->dec32_blob, bytes);
    if (e != hipSuccess) return set_error(WH_ERR_HIP, "hipMalloc(%zu) for the tiled decoder weights failed: %s", bytes, hipGetErrorString(e));
    WH_HIP(hipMemset(m->dec32_blob, 0, bytes));
    Carver c; c.base = (char*)m->dec32_blob;
    m->dec32.resize(L);
    hipStream_t st = nullptr;
    for (size_t l = 0; l < L; ++l) {
        const DecLayerW& w = m->dec[l];
        Dec32LayerW& t = m->dec32[l];
        f16* p;
        p = c.take<f16>(3 * d * d); dec32_tile_weights(w.qkv_w, 3 * d, d, p, st); t.qkv_t = p;
        p = c.take<f16>(d * d); dec32_tile_weights(w.o_w, d, d, p, st); t.o_t = p;
        p = c.take<f16>(d * d); dec32_tile_weights(w.cq_w, d, d, p, st); t.cq_t = p;
        p = c.take<f16>(d * d); dec32_tile_weights(w.co_w, d, d, p, st); t.co_t = p;
        p = c.take<f16>(4 * d * d); dec32_tile_weights(w.fc1_w, 4 * d, d, p, st); t.fc1_t = p;
        p = c.take<f16>(4 * d * d); dec32_tile_weights(w.fc2_w, d, 4 * d, p, st); t.fc2_t = p;
        float *g, *cc;
        g = c.take<float>(3 * d); cc = c.take<float>(3 * d); dec32_fold_vectors(w.qkv_w, 3 * d, d, w.ln1_g, w.ln1_b, w.qkv_b, g, cc, st); t.qkv_g = g; t.qkv_c = cc;
        g = c.take<float>(d); cc = c.take<float>(d); dec32_fold_vectors(w.cq_w, d, d, w.ln2_g, w.ln2_b, w.cq_b, g, cc, st); t.cq_g = g; t.cq_c = cc;
        g = c.take<float>(4 * d); cc = c.take<float>(4 * d); dec32_fold_vectors(w.fc1_w, 4 * d, d, w.ln3_g, w.ln3_b, w.fc1_b, g, cc, st); t.fc1_g = g; t.fc1_c = cc;
    }
    f16* et = c.take<f16>(Vp * d); dec32_tile_weights(m->emb, V, d, et, st); m->emb_t = et;
    float* g = c.take<float>(Vp); float* cc = c.take<float>(Vp);
    dec32_fold_vectors(m->emb, V, d, m->lnf_g, m->lnf_b, nullptr, g, cc, st); m->lg_g = g; m->lg_c = cc;
    if (c.off > bytes) return set_error(WH_ERR_HIP, "internal: tiled decoder weights overflow (%zu > %zu)", c.off, bytes);
    WH_CHECK_LAUNCH();
    WH_HIP(hipDeviceSynchronize());
    return WH_OK;
}

// Absorbed cross-attention weights (xabs.hip): per layer W_k^T as A-fragment tiles of the Q' projection and W_v in the decoder
// projection tiling; b_v stays where it is in the blob.
// Built lazily, by the first session that uses the absorbed path (under the model's lock): a model that only ever carries K / V-row
// sessions (small batches, beam search, WH_XABS=0) does not pay the 2 L d^2 bytes (210 MB at large-v3) per device copy.
static int build_xabs(wh_model* m) {
    const wh_dims& D = m->dims;
    const size_t d = D.n_text_state, L = D.n_text_layer, H = D.n_text_head;
    if (!xabs_supported((int)d, (int)H)) return WH_OK;
    std::lock_guard<std::mutex> lk(m->xabs_mu);
    if (!m->xabs.empty()) return WH_OK;
    WH_HIP(hipSetDevice(m->device));
    const size_t bytes = L * (2 * d * d * 2 + 2 * 256);
    hipError_t e = hipMalloc(&m->xabs_blob, bytes);
    if (e != hipSuccess) return set_error(WH_ERR_HIP, "hipMalloc(%zu) for the absorbed cross-attention weights failed: %s", bytes, hipGetErrorString(e));
    Carver c; c.base = (char*)m->xabs_blob;
    std::vector<wh::XabsLayerW> tiles(L);
    // a private non-blocking stream: the build may run while other sessions of this model are decoding, and a device-wide synchronisation
    // on the null stream would wait for all of their work with the model's lock held
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { hipFree(m->xabs_blob); m->xabs_blob = nullptr; return set_error(WH_ERR_HIP, "stream for the absorbed cross-attention weights"); }
    for (size_t l = 0; l < L; ++l) {
        const f16* wk = m->ckv_w + (l * 2 * d) * d;
        const f16* wv = m->ckv_w + (l * 2 * d + d) * d;
        f16* p = c.take<f16>(d * d); xabs_tile_wk(wk, (int)d, (int)H, p, st); tiles[l].wkT = p;
        p = c.take<f16>(d * d); dec32_tile_weights(wv, (int)d, (int)d, p, st); tiles[l].wv_t = p;
        tiles[l].bv = m->ckv_b + l * 2 * d + d;
    }
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) le = hipStreamSynchronize(st);
    hipStreamDestroy(st);
    if (le != hipSuccess) {         // nothing half-built stays behind: the next absorbed session tries again
        hipFree(m->xabs_blob);
        m->xabs_blob = nullptr;
        return set_error(WH_ERR_HIP, "building the absorbed cross-attention weights failed: %s", hipGetErrorString(le));
    }
    m->xabs.swap(tiles);          // published complete: sessions only read it after their own build_xabs call returned
    return WH_OK;
}

static int set_alignment_heads(wh_model* m, const int32_t* pairs, int n) {
    const int L = m->dims.n_text_layer, H = m->dims.n_text_head;
    std::vector<int> slot(L * H, -1);
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        int l = pairs[2 * i], h = pairs[2 * i + 1];
        if (l < 0 || l >= L || h < 0 || h >= H) return set_error(WH_ERR_INVALID_ARGUMENT, "alignment head (%d,%d) out of range", l, h);
        if (slot[l * H + h] < 0) slot[l * H + h] = cnt++;
    }
    m->align_slot = slot;
    m->n_align = cnt;
    if (!m->align_slot_dev) WH_HIP(hipMalloc((void**)&m->align_slot_dev, sizeof(int) * L * H));
    WH_HIP(hipMemcpy(m->align_slot_dev, slot.data(), sizeof(int) * L * H, hipMemcpyHostToDevice));
    return WH_OK;
}

static int model_create_impl(const void* blob, size_t nbytes, int device, wh_model** out);
// C entry points never let a C++ exception (bad_alloc from a hostile header, ...) cross the ABI
extern "C" int wh_model_create(const void* blob, size_t nbytes, int device, wh_model** out) {
    try { return model_create_impl(blob, nbytes, device, out); }
    catch (const std::exception& e) { return set_error(WH_ERR_MODELS_UNAVAILABLE, "wh_model_create: %s", e.what()); }
    catch (...) { return set_error(WH_ERR_MODELS_UNAVAILABLE, "wh_model_create: unknown exception"); }
}
static int model_create_impl(const void* blob, size_t nbytes, int device, wh_model** out) {
    if (!blob || !out || nbytes < 56) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_model_create: null or truncated blob");
    const unsigned char* p = (const unsigned char*)blob;
    if (memcmp(p, "WHIPW001", 8) != 0) return set_error(WH_ERR_MODELS_UNAVAILABLE, "wh_model_create: bad magic (expected WHIPW001)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(WH_ERR_HIP, "no HIP device visible: the whisperhip product path has no CPU fallback");
    if (device < 0 || device >= ndev) return set_error(WH_ERR_INVALID_ARGUMENT, "device %d out of range (%d visible)", device, ndev);
    WH_HIP(hipSetDevice(device));
    wh_model* m = new wh_model();
    m->device = device;
    memcpy(&m->dims, p + 8, sizeof(wh_dims));
    int32_t n_tensors;
    memcpy(&n_tensors, p + 48, 4);
    const wh_dims& D = m->dims;
    if (D.n_audio_ctx != kCtx || D.n_audio_state < 64 || D.n_audio_state % 64 || D.n_audio_head < 1 || D.n_text_head < 1 ||
        D.n_audio_state / D.n_audio_head != 64 || D.n_audio_state % D.n_audio_head || D.n_text_state / D.n_text_head != 64 || D.n_text_state % D.n_text_head ||
        D.n_audio_state != D.n_text_state || D.n_audio_state > 1280 || (D.n_mels != 80 && D.n_mels != 128) ||
        D.n_audio_layer < 1 || D.n_audio_layer > 64 || D.n_text_layer < 1 || D.n_text_layer > 64 || D.n_vocab < 51864 || D.n_vocab > kMaxVocab ||
        D.n_text_ctx < kMaxTok || D.n_text_ctx > 4096 || n_tensors <= 0 || n_tensors > 65536 ||
        56 + (size_t)n_tensors * sizeof(BlobEntry) > nbytes) {
        delete m;
        return set_error(WH_ERR_MODELS_UNAVAILABLE, "unsupported model dimensions in blob header");
    }
    hipError_t e = hipMalloc(&m->blob_dev, nbytes);
    if (e != hipSuccess) { delete m; return set_error(WH_ERR_HIP, "hipMalloc(%zu) for weights failed: %s", nbytes, hipGetErrorString(e)); }
    m->blob_bytes = nbytes;
    e = hipMemcpy(m->blob_dev, blob, nbytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { wh_model_destroy(m); return set_error(WH_ERR_HIP, "weight upload failed: %s", hipGetErrorString(e)); }
    for (int i = 0; i < n_tensors; ++i) {
        BlobEntry en;
        memcpy(&en, p + 56 + (size_t)i * sizeof(BlobEntry), sizeof(BlobEntry));
        if (en.offset < 0 || en.nbytes < 0 || (size_t)en.offset > nbytes || (size_t)en.nbytes > nbytes - (size_t)en.offset) {
            wh_model_destroy(m);
            return set_error(WH_ERR_MODELS_UNAVAILABLE, "tensor %d out of blob bounds", i);
        }
        WhTensor t;
        t.dev = (char*)m->blob_dev + en.offset;
        t.dtype = en.dtype; t.ndim = en.ndim; t.nbytes = (size_t)en.nbytes;
        for (int k = 0; k < 4; ++k) t.shape[k] = en.shape[k];
        char nm[65]; memcpy(nm, en.name, 64); nm[64] = 0;
        m->t[nm] = t;
    }
    int r = bind_weights(m);
    if (!r) r = build_mel_tables(m);
    if (!r) r = build_dec32(m);
    if (!r) {
        std::vector<int32_t> pairs;
        auto ah = m->t.find("dec.alignment_heads");      // optional int32 [n][2] written by checkpoint conversion (generation_config.alignment_heads)
        if (ah != m->t.end() && ah->second.dtype == 2 && ah->second.nbytes % 8 == 0 && ah->second.nbytes > 0) {
            pairs.resize(ah->second.nbytes / 4);
            if (hipMemcpy(pairs.data(), ah->second.dev, ah->second.nbytes, hipMemcpyDeviceToHost) != hipSuccess) pairs.clear();
        }
        if (pairs.empty())            // default: every head of the upper half of the decoder (openai/whisper model.py)
            for (int l = D.n_text_layer / 2; l < D.n_text_layer; ++l)
                for (int h = 0; h < D.n_text_head; ++h) { pairs.push_back(l); pairs.push_back(h); }
        r = set_alignment_heads(m, pairs.data(), (int)pairs.size() / 2);
    }
    if (r) { wh_model_destroy(m); return r; }
    if (hipMalloc((void**)&m->xattn_gate, 256) != hipSuccess || hipMemset(m->xattn_gate, 0, 256) != hipSuccess) {      // its own cache lines
        wh_model_destroy(m);
        return set_error(WH_ERR_HIP, "hipMalloc of the cross-attention gate word failed");
    }
    *out = m;
    return WH_OK;
}

extern "C" int wh_model_load(const char* path, int device, wh_model** out) {
    if (!path) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_model_load: null path");
    FILE* f = fopen(path, "rb");
    if (!f) return set_error(WH_ERR_MODELS_UNAVAILABLE, "cannot open model file '%s'", path);
    long n = -1;
    if (fseek(f, 0, SEEK_END) == 0) n = ftell(f);
    if (n < 56 || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return set_error(WH_ERR_MODELS_UNAVAILABLE, "cannot size model file '%s'", path); }
    try {
        std::vector<unsigned char> buf((size_t)n);
        size_t got = fread(buf.data(), 1, (size_t)n, f);
        fclose(f);
        if (got != (size_t)n) return set_error(WH_ERR_MODELS_UNAVAILABLE, "short read on '%s'", path);
        return wh_model_create(buf.data(), buf.size(), device, out);
    } catch (const std::exception& e) {
        fclose(f);
        return set_error(WH_ERR_MODELS_UNAVAILABLE, "wh_model_load('%s'): %s", path, e.what());
    }
}

extern "C" void wh_model_destroy(wh_model* m) {
    if (!m) return;
    if (m->blob_dev) hipFree(m->blob_dev);
    if (m->dec32_blob) hipFree(m->dec32_blob);
    if (m->xabs_blob) hipFree(m->xabs_blob);
    if (m->mel_tables_dev) hipFree(m->mel_tables_dev);
    if (m->align_slot_dev) hipFree(m->align_slot_dev);
    if (m->xattn_gate) hipFree(m->xattn_gate);
    delete m;
}

extern "C" int wh_model_dims(const wh_model* m, wh_dims* out) {
    if (!m || !out) return set_error(WH_ERR_MODELS_UNAVAILABLE, "model is null");
    *out = m->dims;
    return WH_OK;
}
extern "C" int wh_model_set_alignment_heads(wh_model* m, const int32_t* pairs, int n) {
    if (!m || (!pairs && n > 0)) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_model_set_alignment_heads: null argument");
    return set_alignment_heads(m, pairs, n);
}
extern "C" int wh_mel_count(const wh_model* m) { return m ? m->dims.n_mels : -1; }
extern "C" int wh_window_samples(const wh_model* m) { return m ? kWindowSamples : -1; }
extern "C" int wh_embed_size(const wh_model* m) { return m ? m->dims.n_audio_state : -1; }
extern "C" int wh_logits_size(const wh_model* m) { return m ? m->dims.n_vocab : -1; }
extern "C" int wh_kv_cache_embed_dim(const wh_model* m) { return m ? m->dims.n_text_state * m->dims.n_text_layer : -1; }
extern "C" int wh_kv_cache_max_sequence_length(const wh_model* m) { return m ? kMaxTok : -1; }
extern "C" int wh_window_size(const wh_model* m) { return m ? m->dims.n_audio_ctx : -1; }
extern "C" int wh_is_model_multilingual(const wh_model* m) { return m ? (m->dims.n_vocab >= 51865) : -1; }
extern "C" int wh_supports_word_timestamps(const wh_model* m) { return m ? (m->n_align > 0) : -1; }

extern "C" int wh_special_tokens_default(const wh_model* m, wh_special_tokens* o) {
    if (!m || !o) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_special_tokens_default: null argument");
    int V = m->dims.n_vocab, eot, nlang;
    if (V == 51864) { eot = 50256; nlang = 99; }
    else if (V == 51865) { eot = 50257; nlang = 99; }
    else if (V == 51866) { eot = 50257; nlang = 100; }
    else return set_error(WH_ERR_TOKENIZER_UNAVAILABLE, "no default special tokens for vocabulary size %d", V);
    int sot = eot + 1, lang0 = sot + 1, translate = lang0 + nlang;
    o->end_token = eot; o->english_token = lang0; o->no_speech_token = translate + 4; o->no_timestamps_token = translate + 5;
    o->special_token_begin = eot; o->start_of_previous_token = translate + 3; o->start_of_transcript_token = sot;
    o->time_token_begin = translate + 6; o->transcribe_token = translate + 1; o->translate_token = translate; o->whitespace_token = 220;
    o->language_token_begin = lang0; o->n_language_tokens = nlang;
    return WH_OK;
}

extern "C" void wh_decoding_options_default(wh_decoding_options* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->task = 0; o->language_token = -1; o->temperature = 0.0f; o->temperature_increment_on_fallback = 0.2f;
    o->temperature_fallback_count = 5; o->sample_length = WH_MAX_TOKEN_CONTEXT; o->top_k = 5; o->use_prefill_prompt = 1;
    o->detect_language = -1; o->skip_special_tokens = 0; o->without_timestamps = 0; o->word_timestamps = 0;
    o->max_initial_timestamp = NAN; o->max_window_seek = -1; o->clip_timestamps = nullptr; o->n_clip_timestamps = 0;
    o->window_clip_time = 1.0f; o->prompt_tokens = nullptr; o->n_prompt_tokens = 0; o->prefix_tokens = nullptr; o->n_prefix_tokens = 0;
    o->suppress_blank = 0; o->suppress_tokens = nullptr; o->n_suppress_tokens = 0;
    o->compression_ratio_threshold = 2.4f; o->log_prob_threshold = -1.0f; o->first_token_log_prob_threshold = -1.5f;
    o->no_speech_threshold = 0.6f; o->seed = 0; o->float16_logits = 0; o->beam_size = 0; o->beam_patience = 1.0f;
}

// ------------------------------------------------------------------------------------------------ session
template <typename T>
static int dalloc(T** p, size_t n, bool zero = true) {
    WH_HIP(hipMalloc((void**)p, n * sizeof(T)));
    if (zero) WH_HIP(hipMemset(*p, 0, n * sizeof(T)));
    return WH_OK;
}
#define DALLOC(ptr, n) do { int _r = dalloc(&(ptr), (size_t)(n)); if (_r) { wh_session_destroy(s); return _r; } } while (0)

static int session_create_impl(wh_model* m, int max_batch, int cross_attention_mode, int cross_attention_splits, int slots_per_workgroup, wh_session** out);
extern "C" int wh_session_create(wh_model* m, int max_batch, wh_session** out) { return session_create_impl(m, max_batch, -1, 0, 0, out); }
extern "C" int wh_session_create_tuned(wh_model* m, int max_batch, int cross_attention_mode, int cross_attention_splits, wh_session** out) {
    wh_session_options o{};
    o.cross_attention_mode = cross_attention_mode; o.cross_attention_splits = cross_attention_splits;
    return wh_session_create_with_options(m, max_batch, &o, out);
}
extern "C" void wh_session_options_default(wh_session_options* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->cross_attention_mode = -1;
}
extern "C" int wh_session_create_with_options(wh_model* m, int max_batch, const wh_session_options* opt, wh_session** out) {
    wh_session_options dflt;
    wh_session_options_default(&dflt);
    if (!opt) opt = &dflt;
    const int cross_attention_mode = opt->cross_attention_mode, cross_attention_splits = opt->cross_attention_splits;
    if (opt->cross_attention_slots_per_workgroup < 0 || opt->cross_attention_slots_per_workgroup > kXabsMaxSlotsPerWorkgroup)
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_session_create: cross_attention_slots_per_workgroup %d (expected 0 auto, 1 .. %d)", opt->cross_attention_slots_per_workgroup, kXabsMaxSlotsPerWorkgroup);
    if (cross_attention_mode < -1 || cross_attention_mode > 1)
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_session_create: cross_attention_mode %d (expected -1 auto, 0 K / V rows, 1 absorbed)", cross_attention_mode);
    if (cross_attention_splits < 0 || cross_attention_splits > kXabsSplits)
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_session_create: cross_attention_splits %d (expected 0 auto, 1 .. %d)", cross_attention_splits, kXabsSplits);
    if (cross_attention_mode == 1 && m && !xabs_supported(m->dims.n_text_state, m->dims.n_text_head))
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_session_create: the absorbed cross-attention needs a model width of 512 / 768 / 1024 / 1280 (this model: %d)", m->dims.n_text_state);
    return session_create_impl(m, max_batch, cross_attention_mode, cross_attention_splits, opt->cross_attention_slots_per_workgroup, out);
}
extern "C" int wh_session_create_with_mode(wh_model* m, int max_batch, int cross_attention_mode, wh_session** out) {
    return wh_session_create_tuned(m, max_batch, cross_attention_mode, 0, out);
}

// Session stream.  Default: a non-blocking stream on the whole chip.  WH_CU_PARTS = N (experiment, round 6): the sessions of a process take turns at N
// compute-unit partitions - session k's stream carries the CU mask of bits [k' W, k' W + W + WH_CU_PART_EXTRA), k' = k % N, W = CUs / N - so that the kernels of
// sessions in flight never wait for each other's workgroups (hipExtStreamCreateWithCUMask: "the first 32 bits represent the first 32 CUs").
static hipError_t create_session_stream(hipStream_t* st) {
    static std::atomic<int> counter{0};
    const char* e = getenv("WH_CU_PARTS");
    const int parts = e ? atoi(e) : 0;
    // WH_STREAM_PRIORITIES = "p0,p1,..." (experiment, round 6): session k's stream gets hardware-queue priority p[k % n] (-1 high, 0 normal, 1 low) - a strict order
    // between the sessions in flight instead of the queues' round robin when their workgroups compete for the same CUs.
    if (const char* pr = getenv("WH_STREAM_PRIORITIES")) {
        int p[8], n = 0;
        for (const char* c = pr; *c && n < 8;) { p[n++] = atoi(c); while (*c && *c != ',') ++c; if (*c == ',') ++c; }
        if (n > 0) return hipStreamCreateWithPriority(st, hipStreamNonBlocking, p[counter.fetch_add(1) % n]);
    }
    if (parts < 2) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const char* x = getenv("WH_CU_PART_EXTRA");
    const int extra = x ? atoi(x) : 0, k = counter.fetch_add(1) % parts, w = n_cu / parts;
    uint32_t mask[16] = {};
    for (int i = k * w; i < k * w + w + extra; ++i) { const int c = ((i % n_cu) + n_cu) % n_cu; mask[c >> 5] |= 1u << (c & 31); }
    return hipExtStreamCreateWithCUMask(st, (uint32_t)((n_cu + 31) / 32), mask);
}
static int session_create_impl(wh_model* m, int max_batch, int cross_attention_mode, int cross_attention_splits, int slots_per_workgroup, wh_session** out) {
    if (!m || !out) return set_error(WH_ERR_MODELS_UNAVAILABLE, "wh_session_create: model is null");
    if (max_batch < 1 || max_batch > kMaxSessionSlots) return set_error(WH_ERR_INVALID_ARGUMENT, "max_batch %d out of range [1, %d]", max_batch, kMaxSessionSlots);
    WH_HIP(hipSetDevice(m->device));
    wh_session* s = new wh_session();
    s->B = max_batch;
    const wh_dims& D = m->dims;
    const size_t B = max_batch, d = D.n_audio_state, L = D.n_text_layer, V = D.n_vocab, H = D.n_text_head;
    if (create_session_stream(&s->st) != hipSuccess) { delete s; return set_error(WH_ERR_HIP, "hipStreamCreate failed"); }
    s->m = m;                       // from here on wh_session_destroy undoes the count
    m->n_sessions.fetch_add(1);
    DALLOC(s->pcm, B * kWindowSamples); DALLOC(s->n_valid, B);
    DALLOC(s->logspec, B * D.n_mels * kFrames); DALLOC(s->maxkey, B);
    DALLOC(s->mel_t, B * kFramesPad * D.n_mels); DALLOC(s->mel_f32, B * D.n_mels * kFrames);
    DALLOC(s->h1, B * kFramesPad * d); DALLOC(s->x, B * kCtx * d); DALLOC(s->xn, B * kCtx * d);
    DALLOC(s->q16, B * kCtx * d); DALLOC(s->k16, B * kCtx * d); DALLOC(s->vt16, B * d * kCtxPad); DALLOC(s->att16, B * kCtx * d);
    DALLOC(s->hmlp, B * kCtx * 4 * d); DALLOC(s->enc16, B * kCtx * d); DALLOC(s->enc32, B * kCtx * d);
    // Cross-attention path, fixed per session: the absorbed form (xabs.hip) streams the encoder output instead of per-layer K / V rows;
    // it pays from about kXabsAutoMinSlots slots (three launches per layer instead of one, one workgroup per (slot, key split) and CU;
    // profiles/r04*, r05*).  Both modes meet the 1e-3 relative logits contract against fp32 (the K / V rows carry 19 mantissa bits since round 5: kernels.h hr24).
    // WH_XABS=0 / 1 forces the choice (A/B, tests), WH_XABS_MIN_SLOTS moves the automatic threshold.
    {
        const char* e_ = getenv("WH_XABS");       // read per session: a process can hold sessions of both modes (tests, A/B)
        const int xabs_mode = cross_attention_mode >= 0 ? cross_attention_mode : (e_ ? atoi(e_) : -1);
        s->use_xabs = xabs_supported((int)d, (int)H) && (xabs_mode < 0 ? max_batch >= wh_xabs_auto_min_slots() : xabs_mode != 0);
        if (s->use_xabs) { const int r_ = build_xabs(m); if (r_) { wh_session_destroy(s); return r_; } }
    }
    if (s->use_xabs) {
        const size_t nht = H > 16 ? 2 : 1, S = kXabsSplits;
        const size_t bytes = 2 * (B * nht * (d / 32) * 1024) + S * H * (d / 8) * B * 32 + S * H * B * 8 + 4 * 256;
        if (hipMalloc(&s->xabs_blob, bytes) != hipSuccess || hipMemset(s->xabs_blob, 0, bytes) != hipSuccess) {
            wh_session_destroy(s);
            return set_error(WH_ERR_HIP, "hipMalloc(%zu) for the absorbed cross-attention buffers failed", bytes);
        }
        Carver c; c.base = (char*)s->xabs_blob;
        s->xabs.layers_host = m->xabs.data();
        s->xabs.n_split = cross_attention_splits > 0 ? cross_attention_splits : xabs_splits(max_batch);
        s->xabs.spw = slots_per_workgroup > 0 ? slots_per_workgroup : 1;
        { const char* e = getenv("WH_XABS_SPW"); const int v = e ? atoi(e) : 0; if (v >= 1 && v <= kXabsMaxSlotsPerWorkgroup) s->xabs.spw = v; }      // A/B override
        s->xabs.qf_hi = c.take<f16>(B * nht * (d / 32) * 512); s->xabs.qf_lo = c.take<f16>(B * nht * (d / 32) * 512);
        s->xabs.part = c.take<float>(S * H * (d / 8) * B * 8);
        s->xabs.ml = c.take<float2>(S * H * B);
    } else {
        DALLOC(s->cross_k_hi, L * B * kCtx * d); DALLOC(s->cross_v_hi, L * B * kCtx * d);
        DALLOC(s->cross_k_lo, L * B * kCtx * d); DALLOC(s->cross_v_lo, L * B * kCtx * d);
    }
    DALLOC(s->self_k, L * B * kMaxTok * d); DALLOC(s->self_v, L * B * kMaxTok * d);
    DALLOC(s->part, B * H * kMaxSplit * kPartStride); DALLOC(s->ticket, B * H);
    DALLOC(s->logits, B * V);
    DALLOC(s->align_mean, B * kMaxTok * kCtx);
    DALLOC(s->seq, B); DALLOC(s->cfg_dev, 1); DALLOC(s->suppress_dev, kMaxSuppress); DALLOC(s->sup_mask_dev, V); DALLOC(s->stats, B * kStatBlocks * 8);
    DALLOC(s->tok_out_dev, B); DALLOC(s->lp_out_dev, B); DALLOC(s->scratch_logits, V);
    {
        const size_t n_bt = (B + 31) / 32, R = n_bt * 32;
        const size_t bytes = 2 * R * d * 4 + 4 * R * d * 2 + 2 * R * 4 * d * 2 + n_bt * (d / 32) * 32 * 8 + n_bt * (size_t)kD32PartFloats * 4 + n_bt * 4096 * 4 + 16 * 256;
        if (hipMalloc(&s->d32_blob, bytes) != hipSuccess || hipMemset(s->d32_blob, 0, bytes) != hipSuccess) {
            wh_session_destroy(s);
            return set_error(WH_ERR_HIP, "hipMalloc(%zu) for the decode-step buffers failed", bytes);
        }
        Carver c; c.base = (char*)s->d32_blob;
        Dec32& q = s->d32;
        q.layers_host = m->dec32.data(); q.emb_t = m->emb_t; q.lg_g = m->lg_g; q.lg_c = m->lg_c; q.n_bt = (int)n_bt;
        q.x = c.take<float>(R * d); q.q = c.take<float>(R * d);
        q.za_hi = c.take<f16>(R * d); q.za_lo = c.take<f16>(R * d); q.zb_hi = c.take<f16>(R * d); q.zb_lo = c.take<f16>(R * d);
        q.h = c.take<f16>(R * 4 * d); q.h_lo = c.take<f16>(R * 4 * d);
        q.stat = c.take<float2>(n_bt * (d / 32) * 32);
        q.part = c.take<float>(n_bt * (size_t)kD32PartFloats); q.part_floats = kD32PartFloats;
        q.ticket = c.take<int>(n_bt * 4096);
    }
    if (hipHostMalloc((void**)&s->seq_host, sizeof(SeqState) * B) != hipSuccess) { wh_session_destroy(s); return set_error(WH_ERR_HIP, "hipHostMalloc failed"); }
    for (auto& e : s->ev) hipEventCreate(&e);
    (void)wh::debug_buffer();   // WH_DBG=1 probe buffer must exist before any stream capture
    // the zero-fills above ran on the NULL stream, which does not order against the session's non-blocking stream: make sure
    // they are done before the first kernel (a late memset of e.g. the arrival counters would corrupt a running decode)
    if (hipDeviceSynchronize() != hipSuccess) { wh_session_destroy(s); return set_error(WH_ERR_HIP, "hipDeviceSynchronize failed after session allocation"); }
    *out = s;
    return WH_OK;
}

extern "C" void wh_session_destroy(wh_session* s) {
    if (!s) return;
    if (s->m) s->m->n_sessions.fetch_sub(1);
    if (s->st) hipStreamSynchronize(s->st);
    whi::drop_session_graphs(s);
    if (s->d32_blob) hipFree(s->d32_blob);
    if (s->xabs_blob) hipFree(s->xabs_blob);
    if (s->align_tmp) hipFree(s->align_tmp);
    void* ptrs[] = {s->pcm, s->n_valid, s->logspec, s->maxkey, s->mel_t, s->mel_f32, s->h1, s->x, s->xn, s->q16, s->k16, s->vt16, s->att16,
                    s->hmlp, s->enc16, s->enc32, s->cross_k_hi, s->cross_v_hi, s->cross_k_lo, s->cross_v_lo, s->self_k, s->self_v, s->part, s->ticket, s->logits,
                    s->align, s->align_mean, s->seq, s->cfg_dev, s->suppress_dev, s->sup_mask_dev, s->stats, s->tok_out_dev, s->lp_out_dev, s->scratch_logits,
                    s->beam_owner, s->beam_tok, s->beam_lp};
    for (void* p : ptrs) if (p) hipFree(p);
    if (s->seq_host) hipHostFree(s->seq_host);
    for (auto& e : s->ev) if (e) hipEventDestroy(e);
    if (s->st) hipStreamDestroy(s->st);
    delete s;
}
extern "C" int wh_session_max_batch(const wh_session* s) { return s ? s->B : -1; }
extern "C" int wh_session_cross_attention_mode(const wh_session* s) { return s ? (s->use_xabs ? 1 : 0) : -1; }
extern "C" int wh_session_step_graph_count(const wh_session* s) { return s ? (int)s->graphs.size() : -1; }
extern "C" int wh_session_cross_attention_splits(const wh_session* s) { return s ? (s->use_xabs ? s->xabs.n_split : 0) : -1; }
extern "C" int wh_session_cross_attention_slots_per_workgroup(const wh_session* s) { return s ? (s->use_xabs ? s->xabs.spw : 0) : -1; }
// key splits per slot an absorbed session of max_batch slots gets when the caller asks for none (xabs.hip xabs_auto_splits)
extern "C" int wh_xabs_auto_splits(int max_batch) { return wh::xabs_auto_splits(max_batch); }
// slots from which wh_session_create picks the absorbed cross-attention on its own (models whose width supports it)
extern "C" int wh_xabs_auto_min_slots(void) {
    const char* e = getenv("WH_XABS_MIN_SLOTS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : wh::kXabsAutoMinSlots;
}
// Development aid: copy the first `nbytes` of a named decode-step buffer to the host (after the session's stream has drained).
extern "C" int wh_debug_peek(wh_session* s, const char* name, void* out, size_t nbytes) {
    CHECK_SESSION(s);
    if (!name || !out) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_debug_peek: null argument");
    const std::string n(name);
    const void* src = nullptr;
    size_t have = 0;          // bytes the named buffer holds: a request beyond it is refused, never read out of bounds
    const size_t R = (size_t)s->d32.n_bt * 32, d = (size_t)s->m->dims.n_text_state, H = (size_t)s->m->dims.n_text_head, B = (size_t)s->B;
    const size_t nht = H > 16 ? 2 : 1;
    if (n == "x") { src = s->d32.x; have = R * d * 4; } else if (n == "q") { src = s->d32.q; have = R * d * 4; }
    else if (n == "za_hi") { src = s->d32.za_hi; have = R * d * 2; } else if (n == "za_lo") { src = s->d32.za_lo; have = R * d * 2; }
    else if (n == "zb_hi") { src = s->d32.zb_hi; have = R * d * 2; } else if (n == "zb_lo") { src = s->d32.zb_lo; have = R * d * 2; }
    else if (s->use_xabs && n == "qf_hi") { src = s->xabs.qf_hi; have = B * nht * 16 * d * 2; }
    else if (s->use_xabs && n == "qf_lo") { src = s->xabs.qf_lo; have = B * nht * 16 * d * 2; }
    else if (s->use_xabs && n == "part") { src = s->xabs.part; have = (size_t)kXabsSplits * H * d * B * 4; }
    else if (s->use_xabs && n == "ml") { src = s->xabs.ml; have = (size_t)kXabsSplits * H * B * 8; }
    else if (n == "enc16") { src = s->enc16; have = B * kCtx * d * 2; }
    if (!src) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_debug_peek: no buffer named '%s' in this session", name);
    if (nbytes > have) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_debug_peek: '%s' holds %zu bytes, %zu requested", name, have, nbytes);
    WH_HIP(hipStreamSynchronize(s->st));
    WH_HIP(hipMemcpy(out, src, nbytes, hipMemcpyDeviceToHost));
    return WH_OK;
}
extern "C" int wh_session_synchronize(wh_session* s) {
    if (!s) return set_error(WH_ERR_INVALID_ARGUMENT, "session is null");
    WH_HIP(hipStreamSynchronize(s->st));
    return WH_OK;
}
extern "C" void* wh_session_stream(wh_session* s) { return s ? (void*)s->st : nullptr; }


// ------------------------------------------------------------------------------------------------ audio / mel
static int set_audio_common(wh_session* s, int b, const float* pcm, int n, hipMemcpyKind kind) {
    CHECK_SESSION(s); CHECK_SLOT(s, b);
    if (n < 0 || (!pcm && n > 0)) return set_error(WH_ERR_AUDIO_PROCESSING_FAILED, "wh_set_audio: invalid buffer (n=%d)", n);
    int m = std::min(n, kWindowSamples);   // trim
    float* dst = s->pcm + (size_t)b * kWindowSamples;
    if (m > 0) WH_HIP(hipMemcpyAsync(dst, pcm, sizeof(float) * m, kind, s->st));
    if (m < kWindowSamples) WH_HIP(hipMemsetAsync(dst + m, 0, sizeof(float) * (kWindowSamples - m), s->st));   // pad (vDSP_vclr)
    WH_HIP(hipMemcpyAsync(s->n_valid + b, &m, sizeof(int), hipMemcpyHostToDevice, s->st));
    WH_HIP(hipStreamSynchronize(s->st));   // &m is a stack temporary
    return WH_OK;
}
extern "C" int wh_set_audio(wh_session* s, int b, const float* pcm_host, int n) { return set_audio_common(s, b, pcm_host, n, hipMemcpyHostToDevice); }
extern "C" int wh_set_audio_device(wh_session* s, int b, const float* pcm_dev, int n) { return set_audio_common(s, b, pcm_dev, n, hipMemcpyDeviceToDevice); }

extern "C" int wh_log_mel_spectrogram(wh_session* s, int batch) {
    CHECK_SESSION(s); CHECK_BATCH(s, batch);
    launch_log_mel(s->m->mel, s->pcm, s->n_valid, batch, s->logspec, s->maxkey, s->mel_t, s->mel_f32, s->st);
    WH_CHECK_LAUNCH();
    return WH_OK;
}
extern "C" int wh_get_mel(wh_session* s, int b, float* out) {
    CHECK_SESSION(s); CHECK_SLOT(s, b);
    if (!out) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_get_mel: null output");
    size_t n = (size_t)s->m->dims.n_mels * kFrames;
    WH_HIP(hipMemcpyAsync(out, s->mel_f32 + b * n, n * 4, hipMemcpyDeviceToHost, s->st));
    WH_HIP(hipStreamSynchronize(s->st));
    return WH_OK;
}
extern "C" int wh_set_mel(wh_session* s, int b, const float* mel) {
    CHECK_SESSION(s); CHECK_SLOT(s, b);
    if (!mel) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_set_mel: null input");
    const int nm = s->m->dims.n_mels;
    size_t n = (size_t)nm * kFrames;
    WH_HIP(hipMemcpyAsync(s->mel_f32 + b * n, mel, n * 4, hipMemcpyHostToDevice, s->st));
    launch_mel_import(s->mel_f32 + b * n, nm, 1, s->mel_t + (size_t)b * kFramesPad * nm, s->st);
    WH_CHECK_LAUNCH();
    WH_HIP(hipStreamSynchronize(s->st));
    return WH_OK;
}

// ------------------------------------------------------------------------------------------------ encoder
extern "C" int wh_encode_features(wh_session* s, int batch) {
    CHECK_SESSION(s); CHECK_BATCH(s, batch);
    const wh_model* m = s->m;
    const wh_dims& D = m->dims;
    const int d = D.n_audio_state, nm = D.n_mels, M = batch * kCtx;
    hipStream_t st = s->st;
    GemmArgs g{};
    // conv1 (k3 s1 p1) + GELU as a GEMM over 3 consecutive rows of the padded time-major mel
    g.A = s->mel_t; g.W = m->conv1_w; g.bias = m->conv1_b; g.M = batch * kFrames; g.N = d; g.K = 3 * nm; g.lda = nm;
    g.a_rows_per_batch = kFrames; g.a_batch_stride = (long long)kFramesPad * nm; g.ldc = d; g.out16 = s->h1; g.rows_per_batch_out = kFrames;
    g.prof_kind = KK_CONV1;
    launch_gemm(EPI_CONV1, g, st);
    // conv2 (k3 s2 p1) + GELU + positional embedding -> fp32 residual stream
    g = GemmArgs{};
    g.A = s->h1; g.W = m->conv2_w; g.bias = m->conv2_b; g.M = M; g.N = d; g.K = 3 * d; g.lda = 2 * d;
    g.a_rows_per_batch = kCtx; g.a_batch_stride = (long long)kFramesPad * d; g.ldc = d; g.out32 = s->x; g.pos = m->enc_pos; g.rows_per_batch_out = kCtx;
    g.prof_kind = KK_CONV2;
    launch_gemm(EPI_CONV2, g, st);
    for (int l = 0; l < D.n_audio_layer; ++l) {
        const EncLayerW& w = m->enc[l];
        launch_layernorm(s->x, w.ln1_g, w.ln1_b, M, d, s->xn, nullptr, st);
        g = GemmArgs{};
        g.A = s->xn; g.W = w.qkv_w; g.bias = w.qkv_b; g.M = M; g.N = 3 * d; g.K = d; g.lda = d; g.a_rows_per_batch = M; g.ldc = d;
        g.out16 = s->q16; g.k16 = s->k16; g.vt16 = s->vt16; g.d_model = d; g.rows_per_batch_out = kCtx;
        g.prof_kind = KK_ENC_QKV;
        launch_gemm(EPI_QKV_ENC, g, st);
        launch_encoder_attention(s->q16, s->k16, s->vt16, s->att16, batch, D.n_audio_head, d, st);
        g = GemmArgs{};
        g.A = s->att16; g.W = w.o_w; g.bias = w.o_b; g.M = M; g.N = d; g.K = d; g.lda = d; g.a_rows_per_batch = M; g.ldc = d; g.out32 = s->x;
        g.prof_kind = KK_ENC_O;
        launch_gemm(EPI_RESID_F32, g, st);
        launch_layernorm(s->x, w.ln2_g, w.ln2_b, M, d, s->xn, nullptr, st);
        g = GemmArgs{};
        g.A = s->xn; g.W = w.fc1_w; g.bias = w.fc1_b; g.M = M; g.N = 4 * d; g.K = d; g.lda = d; g.a_rows_per_batch = M; g.ldc = 4 * d; g.out16 = s->hmlp;
        g.prof_kind = KK_ENC_FC1;
        launch_gemm(EPI_GELU_F16, g, st);
        g = GemmArgs{};
        g.A = s->hmlp; g.W = w.fc2_w; g.bias = w.fc2_b; g.M = M; g.N = d; g.K = 4 * d; g.lda = 4 * d; g.a_rows_per_batch = M; g.ldc = d; g.out32 = s->x;
        g.prof_kind = KK_ENC_FC2;
        launch_gemm(EPI_RESID_F32, g, st);
    }
    launch_layernorm(s->x, m->lnp_g, m->lnp_b, M, d, s->enc16, s->enc32, st);
    WH_CHECK_LAUNCH();
    return WH_OK;
}
extern "C" int wh_get_encoder_output(wh_session* s, int b, float* out) {
    CHECK_SESSION(s); CHECK_SLOT(s, b);
    if (!out) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_get_encoder_output: null output");
    size_t n = (size_t)kCtx * s->m->dims.n_audio_state;
    WH_HIP(hipMemcpyAsync(out, s->enc32 + b * n, n * 4, hipMemcpyDeviceToHost, s->st));
    WH_HIP(hipStreamSynchronize(s->st));
    return WH_OK;
}
extern "C" int wh_set_encoder_output(wh_session* s, int b, const float* enc) {
    CHECK_SESSION(s); CHECK_SLOT(s, b);
    if (!enc) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_set_encoder_output: null input");
    size_t n = (size_t)kCtx * s->m->dims.n_audio_state;
    WH_HIP(hipMemcpyAsync(s->enc32 + b * n, enc, n * 4, hipMemcpyHostToDevice, s->st));
    launch_f32_to_f16(s->enc32 + b * n, s->enc16 + b * n, n, s->st);
    WH_CHECK_LAUNCH();
    WH_HIP(hipStreamSynchronize(s->st));
    return WH_OK;
}

// ------------------------------------------------------------------------------------------------ decoder
constexpr bool kXattnGateDefault = false;     // flipped once measured (profiles/r03h_*)
namespace whi {
DecodeBuffers decode_buffers(wh_session* s, int batch, int max_position) {
    // max_position: the largest token_index any live slot can have during the launches built from this description
    const wh_model* m = s->m;
    DecodeBuffers db{};
    db.cross_div = 1;
    db.self_rows = std::min(std::max(max_position, 0), kMaxTok - 1) + 1;
    db.self_owner = nullptr;
    db.batch = batch; db.max_batch = s->B; db.d = m->dims.n_text_state; db.n_head = m->dims.n_text_head; db.n_layer = m->dims.n_text_layer; db.n_vocab = m->dims.n_vocab;
    db.emb = m->emb; db.pos = m->dec_pos; db.layers_host = m->dec.data(); db.lnf_g = m->lnf_g; db.lnf_b = m->lnf_b;
    db.self_k = s->self_k; db.self_v = s->self_v; db.cross_k_hi = s->cross_k_hi; db.cross_v_hi = s->cross_v_hi; db.cross_k_lo = s->cross_k_lo; db.cross_v_lo = s->cross_v_lo;
    db.part = s->part; db.ticket = s->ticket; db.logits = s->logits; db.seq = s->seq;
    db.stats = s->stats; db.sup_mask = s->sup_mask_dev; db.fused_greedy = s->fused_greedy ? 1 : 0;
    db.align = s->align_enabled ? s->align : nullptr; db.align_slot = m->align_slot_dev; db.n_align = s->n_align_alloc;
    db.d32 = &s->d32; db.x = s->d32.x; db.q = s->d32.q;
    if (s->use_xabs) { s->xabs.enc = s->enc16; db.xabs = &s->xabs; }
    // cross-attention gate: WH_XATT_GATE=0 never, 1 always, unset: while the model carries more than one session (dec_shared.h)
    static const int gate_mode = [] { const char* e = getenv("WH_XATT_GATE"); return e ? atoi(e) : -1; }();
    const bool gate_on = gate_mode < 0 ? kXattnGateDefault && m->n_sessions.load() > 1 : gate_mode != 0;
    db.xattn_gate = gate_on ? m->xattn_gate : nullptr;
    return db;
}
}  // namespace whi

namespace whi {
int ensure_align(wh_session* s) {
    // The raw score buffer is sized for the model's alignment-head count at allocation time; wh_model_set_alignment_heads may
    // have changed it since (the captured step graphs bake the row stride in: they are keyed by n_align and dropped here).
    if (s->align && s->n_align_alloc != s->m->n_align) {
        WH_HIP(hipStreamSynchronize(s->st));
        drop_session_graphs(s);
        hipFree(s->align);
        s->align = nullptr;
        s->n_align_alloc = 0;      // nothing is allocated: sizes derived from it (graph keys, align_tmp) must not see the old head count
    }
    if (!s->align && s->m->n_align > 0) {
        size_t n = (size_t)s->B * kMaxTok * s->m->n_align * kCtx;
        WH_HIP(hipMalloc((void**)&s->align, n * sizeof(float)));
        WH_HIP(hipMemsetAsync(s->align, 0, n * sizeof(float), s->st));
        s->n_align_alloc = s->m->n_align;
    }
    return WH_OK;
}

int reset_decoder_inputs_masked(wh_session* s, int batch, const int32_t* active) {
    // DecodingInputs.reset for the slots that decode again (temperature fallback): an accepted slot keeps its alignment rows
    // until the window's word timestamps have been read (TranscribeTask.swift:374-398 resets only the task's own inputs)
    for (int b = 0; b < batch; ++b) {
        if (active && !active[b]) continue;
        WH_HIP(hipMemsetAsync(s->seq + b, 0, sizeof(SeqState), s->st));
        if (s->align) WH_HIP(hipMemsetAsync(s->align + (size_t)b * kMaxTok * s->n_align_alloc * kCtx, 0, (size_t)kMaxTok * s->n_align_alloc * kCtx * sizeof(float), s->st));
    }
    return WH_OK;
}
}  // namespace whi
using whi::ensure_align;

extern "C" int wh_reset_decoder_inputs(wh_session* s, int batch) {
    CHECK_SESSION(s); CHECK_BATCH(s, batch);
    return whi::reset_decoder_inputs_masked(s, batch, nullptr);
}

extern "C" int wh_prepare_decoder_inputs(wh_session* s, int batch) {
    CHECK_SESSION(s); CHECK_BATCH(s, batch);
    const wh_model* m = s->m;
    const int d = m->dims.n_text_state, L = m->dims.n_text_layer;
    if (!s->use_xabs) {     // (the absorbed cross-attention reads the encoder output itself: no per-layer K / V projection)
        GemmArgs g{};
        g.A = s->enc16; g.W = m->ckv_w; g.bias = m->ckv_b; g.M = batch * kCtx; g.N = L * 2 * d; g.K = d; g.lda = d; g.a_rows_per_batch = g.M;
        g.ldc = L * 2 * d; g.kv_k_hi = s->cross_k_hi; g.kv_v_hi = s->cross_v_hi; g.kv_k_lo = s->cross_k_lo; g.kv_v_lo = s->cross_v_lo; g.d_model = d; g.max_batch = s->B;
        g.prof_kind = KK_CROSS_KV;
        launch_gemm(EPI_CROSS_KV, g, s->st);
        WH_CHECK_LAUNCH();
    }
    return wh_reset_decoder_inputs(s, batch);
}

extern "C" int wh_predict_logits(wh_session* s, int batch, const int32_t* tokens, const int32_t* positions, float* logits_out) {
    CHECK_SESSION(s); CHECK_BATCH(s, batch);
    if (!tokens || !positions) return set_error(WH_ERR_DECODING_LOGITS_FAILED, "wh_predict_logits: null tokens/positions");
    const int V = s->m->dims.n_vocab;
    int r = ensure_align(s);
    if (r) return r;
    s->align_enabled = s->align != nullptr;
    for (int b = 0; b < batch; ++b) {
        if (tokens[b] < 0 || tokens[b] >= V || positions[b] < 0 || positions[b] >= kMaxTok)
            return set_error(WH_ERR_DECODING_LOGITS_FAILED, "wh_predict_logits: token %d / position %d out of range", tokens[b], positions[b]);
        SeqState& q = s->seq_host[b];
        memset(&q, 0, sizeof(q));
        q.next_token = tokens[b]; q.token_index = positions[b]; q.active = 1; q.done = 0;
    }
    // only the control fields are refreshed; token history on the device is not used by the bare step
    WH_HIP(hipMemcpyAsync(s->seq, s->seq_host, sizeof(SeqState) * batch, hipMemcpyHostToDevice, s->st));
    DecodeBuffers db = whi::decode_buffers(s, batch, *std::max_element(positions, positions + batch));
    launch_decoder_step(db, nullptr, nullptr, false, s->st);
    WH_CHECK_LAUNCH();
    if (logits_out) WH_HIP(hipMemcpyAsync(logits_out, s->logits, sizeof(float) * (size_t)batch * V, hipMemcpyDeviceToHost, s->st));
    WH_HIP(hipStreamSynchronize(s->st));
    return WH_OK;
}

extern "C" int wh_get_alignment_weights(wh_session* s, int b, float* out) {
    CHECK_SESSION(s); CHECK_SLOT(s, b);
    if (!out) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_get_alignment_weights: null output");
    if (!s->align) return set_error(WH_ERR_SEGMENTING_FAILED, "no alignment weights recorded (run a decode with word timestamps / the step API first)");
    size_t n = (size_t)kMaxTok * kCtx;
    if (s->align_znorm || s->align_median > 1) {
        const size_t H = (size_t)s->n_align_alloc;
        if (s->align_tmp && s->align_tmp_heads != s->n_align_alloc) { WH_HIP(hipStreamSynchronize(s->st)); hipFree(s->align_tmp); s->align_tmp = nullptr; }
        if (!s->align_tmp) {
            WH_HIP(hipMalloc((void**)&s->align_tmp, ((size_t)kMaxTok * H * kCtx + 2 * H * kCtx + 256) * sizeof(float)));
            s->align_tmp_heads = s->n_align_alloc;
        }
        float* stat = s->align_tmp + (size_t)kMaxTok * H * kCtx;
        launch_alignment_postprocess(s->align + (size_t)b * n * H, s->n_align_alloc, s->align_tmp, stat, reinterpret_cast<int*>(stat + 2 * H * kCtx),
                                     s->align_znorm, s->align_median, s->align_mean + b * n, s->st);
    } else {
        launch_alignment_mean(s->align + (size_t)b * n * s->n_align_alloc, 1, s->n_align_alloc, s->align_mean + b * n, s->st);   // this slot only
    }
    WH_CHECK_LAUNCH();
    WH_HIP(hipMemcpyAsync(out, s->align_mean + b * n, n * 4, hipMemcpyDeviceToHost, s->st));
    WH_HIP(hipStreamSynchronize(s->st));
    return WH_OK;
}

// ---- device-resident hand-off (outputs may stay in HBM) -------------------------
extern "C" int wh_get_mel_device(wh_session* s, int b, const float** mel_dev) {
    CHECK_SESSION(s); CHECK_SLOT(s, b);
    if (!mel_dev) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_get_mel_device: null output");
    *mel_dev = s->mel_f32 + (size_t)b * s->m->dims.n_mels * kFrames;
    return WH_OK;
}
extern "C" int wh_get_encoder_output_device(wh_session* s, int b, const float** enc_f32_dev, const void** enc_f16_dev) {
    CHECK_SESSION(s); CHECK_SLOT(s, b);
    const size_t n = (size_t)kCtx * s->m->dims.n_audio_state;
    if (enc_f32_dev) *enc_f32_dev = s->enc32 + b * n;
    if (enc_f16_dev) *enc_f16_dev = s->enc16 + b * n;
    return WH_OK;
}
extern "C" int wh_get_logits_device(wh_session* s, const float** logits_dev) {
    CHECK_SESSION(s);
    if (!logits_dev) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_get_logits_device: null output");
    *logits_dev = s->logits;
    return WH_OK;
}
static void fill_tensor(wh_tensor* t, const void* data, int dtype, int device, int64_t d0, int64_t d1) {
    memset(t, 0, sizeof(*t));
    t->data = const_cast<void*>(data); t->dtype = dtype; t->ndim = 2; t->shape[0] = d0; t->shape[1] = d1; t->device = device;
}
extern "C" int wh_get_mel_tensor(wh_session* s, int b, wh_tensor* out) {
    if (!out) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_get_mel_tensor: null output");
    const float* p = nullptr;
    int r = wh_get_mel_device(s, b, &p);
    if (r) return r;
    fill_tensor(out, p, WH_DTYPE_F32, s->m->device, s->m->dims.n_mels, kFrames);
    return WH_OK;
}
extern "C" int wh_get_encoder_output_tensor(wh_session* s, int b, int dtype, wh_tensor* out) {
    if (!out || (dtype != WH_DTYPE_F32 && dtype != WH_DTYPE_F16)) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_get_encoder_output_tensor: null output or unknown dtype %d", dtype);
    const float* p32 = nullptr; const void* p16 = nullptr;
    int r = wh_get_encoder_output_device(s, b, &p32, &p16);
    if (r) return r;
    fill_tensor(out, dtype == WH_DTYPE_F32 ? (const void*)p32 : p16, dtype, s->m->device, kCtx, s->m->dims.n_audio_state);
    return WH_OK;
}
extern "C" int wh_get_logits_tensor(wh_session* s, wh_tensor* out) {
    if (!out) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_get_logits_tensor: null output");
    const float* p = nullptr;
    int r = wh_get_logits_device(s, &p);
    if (r) return r;
    fill_tensor(out, p, WH_DTYPE_F32, s->m->device, s->B, s->m->dims.n_vocab);
    return WH_OK;
}

extern "C" int wh_session_set_alignment_postprocess(wh_session* s, int z_normalize, int median_filter_width) {
    if (!s) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_session_set_alignment_postprocess: null session");
    if (median_filter_width < 0 || median_filter_width > 15 || (median_filter_width > 1 && median_filter_width % 2 == 0))
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_session_set_alignment_postprocess: the median filter width must be 0 / 1 (off) or odd and <= 15");
    s->align_znorm = z_normalize != 0;
    s->align_median = median_filter_width;
    return WH_OK;
}
extern "C" int wh_session_set_cancel_flag(wh_session* s, const volatile int32_t* flag) {
    if (!s) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_session_set_cancel_flag: null session");
    s->cancel_flag = flag;
    return WH_OK;
}

// ---- filter / sampler KAT entry points ----------------------------------------------------------
static int upload_cfg(wh_session* s, const wh_decoding_options* opt, const wh_special_tokens* st, int prefilled_index,
                      int initial_prompt_index, int language_filter, int n_vocab, uint64_t seed) {
    SamplerCfg c{};
    c.n_vocab = n_vocab;
    c.end_token = st->end_token; c.no_timestamps_token = st->no_timestamps_token; c.time_token_begin = st->time_token_begin;
    c.transcribe_token = st->transcribe_token; c.translate_token = st->translate_token; c.whitespace_token = st->whitespace_token;
    c.is_multilingual = wh_is_model_multilingual(s->m);
    c.suppress_blank = opt ? opt->suppress_blank : 0; c.prefilled_index = prefilled_index;
    c.timestamp_rules = opt ? !opt->without_timestamps : 0; c.initial_prompt_index = initial_prompt_index;
    c.language_filter = language_filter; c.language_token_begin = st->language_token_begin; c.n_language_tokens = st->n_language_tokens;
    c.top_k = opt ? opt->top_k : 5;
    int sample_length = opt ? opt->sample_length : WH_MAX_TOKEN_CONTEXT;
    c.loop_count = std::min(sample_length, kMaxTok - 1);
    c.has_first_token_threshold = opt && !isnan(opt->first_token_log_prob_threshold);
    c.first_token_log_prob_threshold = opt ? opt->first_token_log_prob_threshold : 0.f;
    c.seed = seed;
    c.f16_logits = opt ? (opt->float16_logits != 0) : 0;
    // createLogitsFilters: suppressTokens filtered to ids < specialTokenBegin (TextDecoder.swift:876-879)
    std::vector<int> sup;
    if (opt && opt->suppress_tokens)
        for (int i = 0; i < opt->n_suppress_tokens; ++i)
            if (opt->suppress_tokens[i] < st->special_token_begin && opt->suppress_tokens[i] >= 0) sup.push_back(opt->suppress_tokens[i]);
    if ((int)sup.size() > kMaxSuppress) return set_error(WH_ERR_INVALID_ARGUMENT, "more than %d suppress tokens", kMaxSuppress);
    c.n_suppress = (int)sup.size();
    {
        std::vector<unsigned char> mask((size_t)s->m->dims.n_vocab, 0);
        for (int t : sup) if (t < (int)mask.size()) mask[t] = 1;
        WH_HIP(hipMemcpyAsync(s->sup_mask_dev, mask.data(), mask.size(), hipMemcpyHostToDevice, s->st));
        WH_HIP(hipStreamSynchronize(s->st));   // mask is a stack temporary
    }
    if (!sup.empty()) WH_HIP(hipMemcpyAsync(s->suppress_dev, sup.data(), sizeof(int) * sup.size(), hipMemcpyHostToDevice, s->st));
    WH_HIP(hipMemcpyAsync(s->cfg_dev, &c, sizeof(c), hipMemcpyHostToDevice, s->st));
    WH_HIP(hipStreamSynchronize(s->st));   // c / sup are stack temporaries
    return WH_OK;
}

namespace whi { int upload_sampler_cfg(wh_session* s, const wh_decoding_options* opt, const wh_special_tokens* st, int prefilled_index,
                                       int initial_prompt_index, int language_filter, uint64_t seed) {
    return upload_cfg(s, opt, st, prefilled_index, initial_prompt_index, language_filter, s->m->dims.n_vocab, seed);
} }

extern "C" int wh_filter_logits(wh_session* s, const wh_decoding_options* opt, const wh_special_tokens* st, const int32_t* tokens,
                                int n_tokens, int prefilled_index, int initial_prompt_index, int language_filter,
                                float* logits, int n_logits) {
    CHECK_SESSION(s);
    if (!st || !logits || n_logits < 1 || n_logits > s->m->dims.n_vocab || n_tokens < 0 || n_tokens > kMaxTok || (n_tokens && !tokens))
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_filter_logits: invalid argument");
    int r = upload_cfg(s, opt, st, prefilled_index, initial_prompt_index, language_filter, n_logits, 0);
    if (r) return r;
    SeqState& q = s->seq_host[0];
    memset(&q, 0, sizeof(q));
    for (int i = 0; i < n_tokens; ++i) q.tokens[i] = tokens[i];
    q.n_tokens = n_tokens; q.active = 1;
    WH_HIP(hipMemcpyAsync(s->seq, &q, sizeof(SeqState), hipMemcpyHostToDevice, s->st));
    WH_HIP(hipMemcpyAsync(s->scratch_logits, logits, sizeof(float) * n_logits, hipMemcpyHostToDevice, s->st));
    launch_filter_only(s->cfg_dev, s->suppress_dev, s->seq, s->scratch_logits, n_logits, s->st);
    WH_CHECK_LAUNCH();
    WH_HIP(hipMemcpyAsync(logits, s->scratch_logits, sizeof(float) * n_logits, hipMemcpyDeviceToHost, s->st));
    WH_HIP(hipStreamSynchronize(s->st));
    return WH_OK;
}

extern "C" int wh_sample_token(wh_session* s, const float* logits, int n_logits, float temperature, int top_k, uint64_t seed,
                               int counter, int32_t* token_out, float* logprob_out) {
    CHECK_SESSION(s);
    if (!logits || !token_out || !logprob_out || n_logits < 1 || n_logits > s->m->dims.n_vocab)
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_sample_token: invalid argument");
    wh_decoding_options o;
    wh_decoding_options_default(&o);
    o.top_k = top_k; o.without_timestamps = 1;
    wh_special_tokens st{};
    int r = upload_cfg(s, &o, &st, 0, 0, 0, n_logits, seed);
    if (r) return r;
    SeqState& q = s->seq_host[0];
    memset(&q, 0, sizeof(q));
    q.active = 1; q.temperature = (float)(_Float16)temperature;   // GreedyTokenSampler.temperature is FloatType (TokenSampler.swift:30)
    WH_HIP(hipMemcpyAsync(s->seq, &q, sizeof(SeqState), hipMemcpyHostToDevice, s->st));
    WH_HIP(hipMemcpyAsync(s->scratch_logits, logits, sizeof(float) * n_logits, hipMemcpyHostToDevice, s->st));
    launch_sample_only(s->cfg_dev, s->seq, s->scratch_logits, n_logits, counter, s->tok_out_dev, s->lp_out_dev, s->st);
    WH_CHECK_LAUNCH();
    WH_HIP(hipMemcpyAsync(token_out, s->tok_out_dev, sizeof(int), hipMemcpyDeviceToHost, s->st));
    WH_HIP(hipMemcpyAsync(logprob_out, s->lp_out_dev, sizeof(float), hipMemcpyDeviceToHost, s->st));
    WH_HIP(hipStreamSynchronize(s->st));
    return WH_OK;
}
