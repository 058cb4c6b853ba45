/* A plain-C host for libwhisperhip.so: WhisperKit's `transcribe(audioPath:)` flow through the C ABI alone.
 *
 *   gcc -std=c99 -Iinclude examples/transcribe.c -Lwhisperkit_amd -lwhisperhip -Wl,-rpath,$PWD/whisperkit_amd -o transcribe
 *   ./transcribe model.whipw tokenizer.json talk.wav talk            -> talk.srt, talk.vtt, talk.json   (needs an MI355X)
 *   ./transcribe --selftest tokenizer.json talk.wav                  -> host-only entry points, no GPU needed
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "whisperhip.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != WH_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, wh_last_error()); return 1; } \
    } while (0)

static int selftest(const char* tokenizer_json, const char* wav) {
    wh_tokenizer* tok = NULL;
    CHECK(wh_tokenizer_load(tokenizer_json, &tok));
    wh_special_tokens st;
    CHECK(wh_tokenizer_special_tokens(tok, &st));
    int32_t ids[3] = {st.start_of_transcript_token, st.time_token_begin + 50, st.end_token};
    char text[128];
    int n = wh_tokenizer_decode(tok, ids, 3, 0, text, (int)sizeof text);
    printf("vocab %d, decode -> %s (%d bytes)\n", wh_tokenizer_vocab_size(tok), text, n);
    float* pcm = NULL;
    int n_samples = 0;
    CHECK(wh_load_audio(wav, WH_CHANNEL_SUM, NULL, 0, 0.0, NAN, 0, &pcm, &n_samples));
    wh_decoding_options opt;
    wh_decoding_options_default(&opt);
    int32_t cs[64], ce[64];
    int nc = wh_vad_chunk_all(pcm, n_samples, WH_WINDOW_SAMPLES, &opt, cs, ce, 64);
    char t0[32];
    wh_format_time((float)n_samples / WH_SAMPLE_RATE, 1, ',', t0, (int)sizeof t0);
    printf("audio %d samples = %s, %d chunk(s)\n", n_samples, t0, nc);
    wh_audio_free(pcm);
    wh_tokenizer_destroy(tok);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 4 && !strcmp(argv[1], "--selftest")) return selftest(argv[2], argv[3]);
    if (argc != 5) { fprintf(stderr, "usage: %s model.whipw tokenizer.json audio.wav out-stem | --selftest tokenizer.json audio.wav\n", argv[0]); return 2; }
    wh_model* model = NULL;
    wh_session* session = NULL;
    wh_tokenizer* tok = NULL;
    CHECK(wh_model_load(argv[1], 0, &model));                 /* WhisperKit.loadModels */
    CHECK(wh_session_create(model, 8, &session));             /* 8 windows in flight */
    CHECK(wh_tokenizer_load(argv[2], &tok));                  /* ModelUtilities.loadTokenizer */
    CHECK(wh_session_set_tokenizer(session, tok));
    wh_special_tokens st;
    CHECK(wh_tokenizer_special_tokens(tok, &st));

    float* pcm = NULL;
    int n_samples = 0;
    CHECK(wh_load_audio(argv[3], WH_CHANNEL_SUM, NULL, 0, 0.0, NAN, 0, &pcm, &n_samples));   /* AudioProcessor.loadAudio */

    wh_decoding_options opt;
    wh_decoding_options_default(&opt);
    opt.word_timestamps = 1;
    enum { CAP = 4096 };
    static wh_transcription* chunks[CAP];
    int n_chunks = 0;
    CHECK(wh_transcribe_chunked(session, pcm, n_samples, &opt, &st, chunks, CAP, NULL, &n_chunks));   /* transcribe, .vad chunking */
    wh_transcription* merged = NULL;
    CHECK(wh_merge_transcriptions((const wh_transcription* const*)chunks, n_chunks, NULL, 0, &merged));

    char path[1024];
    snprintf(path, sizeof path, "%s.srt", argv[4]);  CHECK(wh_write_srt(merged, path));
    snprintf(path, sizeof path, "%s.vtt", argv[4]);  CHECK(wh_write_vtt(merged, path));
    snprintf(path, sizeof path, "%s.json", argv[4]); CHECK(wh_write_json(merged, path));
    wh_timings tm;
    CHECK(wh_transcription_timings(merged, &tm));
    printf("%d chunk(s), %d segment(s), %.1f s of audio in %.3f s (speed factor %.1f)\n", n_chunks, wh_transcription_n_segments(merged),
           tm.input_audio_seconds, tm.full_pipeline, tm.input_audio_seconds / tm.full_pipeline);

    for (int i = 0; i < n_chunks; ++i) wh_transcription_free(chunks[i]);
    wh_transcription_free(merged);
    wh_audio_free(pcm);
    wh_tokenizer_destroy(tok);
    wh_session_destroy(session);
    wh_model_destroy(model);
    return 0;
}
