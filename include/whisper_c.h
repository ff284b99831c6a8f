/* whisper_c.h -- flat C mirror of the COM-style API in whisperApi.h, exported by libWhisper.so for FFI callers that cannot
 * consume C++ vtables (Python ctypes, cgo, JNI, N-API). Every function forwards to the iModel / iContext method named in
 * its comment (reference: Whisper/API/iContext.cl.h:23-60); return values are the HRESULTs of those methods.
 * Opaque handles are COM object pointers: release each with whisperc_release. */
#ifndef WHISPER_C_H
#define WHISPER_C_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Whisper::loadModel( path, { GPU, adapter = device }, nullptr, &model ) */
int32_t whisperc_load_model( const char* pathUtf8, int device, void** modelOut );
/* IUnknown::Release */
void whisperc_release( void* unknown );
/* iModel::createContext */
int32_t whisperc_create_context( void* model, void** ctxOut );
/* iModel::getSpecialTokens -> { eot, sot, prev, solm, not, beg, translate, transcribe } */
int32_t whisperc_special_tokens( void* model, int32_t* out8 );
/* iModel::stringFromToken (pointer into the model's vocabulary, valid while the model lives) */
const char* whisperc_token_string( void* model, int token );
/* iModel::isMultilingual: S_OK (0) or S_FALSE (1) */
int32_t whisperc_is_multilingual( void* model );
/* iModel::tokenize: returns the token count (>= 0) or a failed HRESULT */
int32_t whisperc_tokenize( void* model, const char* text, int32_t* out, int cap );
/* iContext::fullDefaultParams( Greedy ) + the given fields, then iContext::runFull on a mono FP32 16 kHz buffer.
 * flags = eFullParamsFlags bits (Translate 1, NoContext 2, SingleSegment 4, PrintSpecial 8 ...). */
int32_t whisperc_run_full( void* ctx, const float* pcm, uint32_t nSamples, const char* language, uint32_t flags, int maxTokens,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx /* < 0 keeps the default 16384 */ );
/* The same with eSamplingStrategy::BeamSearch and beam_search.beam_width = beamWidth (1 .. 8; the reference declares the strategy,
 * Whisper/API/sFullParams.h:10-13, and implements only Greedy): beamWidth hypotheses per window share one pass over its cross-attention K/V. */
/* whisperc_run_full + sFullParams::audio_ctx (encoder positions / cross-attention keys per window; 0 = the model's) */
int32_t whisperc_run_full_audio_ctx( void* ctx, const float* pcm, uint32_t nSamples, const char* language, uint32_t flags, int maxTokens,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx, int audioCtx );
int32_t whisperc_run_full_beam( void* ctx, const float* pcm, uint32_t nSamples, const char* language, uint32_t flags, int maxTokens,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx, int beamWidth );
/* iMediaFoundation::loadAudioFileData( WAV bytes: 16 kHz, mono/stereo, PCM16/float32 ) + iContext::runStreamed( params,
 * { progress callback }, reader ): the streaming entry the reference's CLI uses by default (Examples/main/main.cpp:305-311).
 * The values the progress sink received are copied to progressOut (first progressCap of them), their count to *progressCount. */
int32_t whisperc_run_streamed( void* ctx, const void* wavBytes, uint64_t wavSize, const char* language, uint32_t flags, int maxTokens,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx, double* progressOut, int progressCap, int* progressCount );
/* whisperc_run_full + the token-timestamp fields of sFullParams (set TokenTimestamps = 0x100 in flags): thold_pt, thold_ptsum, max_len */
int32_t whisperc_run_full_tt( void* ctx, const float* pcm, uint32_t nSamples, const char* language, uint32_t flags, int maxTokens,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx, float tholdPt, float tholdPtsum, int maxLen );
/* iContext::getResults( Tokens | Timestamps ) + iTranscribeResult::getSize / getSegments / getTokens; times in 100 ns ticks */
int32_t whisperc_result_counts( void* ctx, uint32_t* segments, uint32_t* tokens );
int32_t whisperc_result_segment( void* ctx, uint32_t index, uint64_t* t0, uint64_t* t1, uint32_t* firstToken, uint32_t* countTokens,
	char* text, uint32_t textCap );
int32_t whisperc_result_token( void* ctx, uint32_t index, int32_t* id, float* p, float* pt, float* ptsum );
/* sToken::time (100 ns ticks; 0 when unknown) and sToken::vlen of token `index` */
int32_t whisperc_result_token_times( void* ctx, uint32_t index, uint64_t* t0, uint64_t* t1, float* vlen );
/* Whisper::createBatchRunner( model, { maxSlots, groups, greedyChunk, flags } ) (0 = the defaults) -> iBatchRunner; release with whisperc_release */
int32_t whisperc_batch_create( void* model, uint32_t maxSlots, uint32_t groups, uint32_t greedyChunk, uint32_t flags, void** runnerOut );
/* iBatchRunner::run over `count` streams: stream i = samples [firstSample[i], firstSample[i] + countSamples[i]) (0 = to the end) of the mono FP32
 * 16 kHz buffer pcm[i] of nSamples[i] samples (streams may name the same buffer: the chunks of one recording); the common sFullParams are
 * fullDefaultParams( Greedy ) + the given fields, like whisperc_run_full. resultsOut[i] receives an iTranscribeResult (or NULL; release each with
 * whisperc_release), perStream[i] the stream's own HRESULT; both HOST arrays of `count` entries, perStream may be NULL. */
int32_t whisperc_batch_run( void* runner, uint32_t count, const float* const* pcm, const uint32_t* nSamples, const int64_t* firstSample,
	const int64_t* countSamples, const char* language, uint32_t flags, int maxTokens, const int32_t* promptTokens, int nPrompt, int nMaxTextCtx,
	void** resultsOut, int32_t* perStream );
/* iTranscribeResult::getSize / getSegments / getTokens on a result object itself (times in 100 ns ticks) */
int32_t whisperc_tr_counts( void* result, uint32_t* segments, uint32_t* tokens );
int32_t whisperc_tr_segment( void* result, uint32_t index, uint64_t* t0, uint64_t* t1, uint32_t* firstToken, uint32_t* countTokens, char* text, uint32_t textCap );
int32_t whisperc_tr_token( void* result, uint32_t index, int32_t* id, float* p, float* pt, float* ptsum, uint64_t* t0, uint64_t* t1, float* vlen );
/* iContext::timingsPrint */
int32_t whisperc_timings_print( void* ctx );
/* One line of the profiler output, formatted like ProfileCollection::Measure::print (Whisper/Utils/ProfileCollection.cpp:113-170):
 * `ticks` of 100 ns scaled to seconds / milliseconds / microseconds. Returns the length written (without the terminator). */
int32_t whisperc_format_measure( const char* name, double ticks, uint64_t count, char* out, uint32_t outCap );
/* The TokenTimestamps post-processing on its own (host only, no device): token data of finished segments in, token times
 * (10 ms units) and -- with maxLen > 0 -- the segments wrapped to maxLen characters out; iContext::runFull applies exactly
 * this per segment (whisper.cpp:3374-3575, 2711-2760). segTimes / outSegTimes / outTokTimes hold (t0, t1) pairs, tokens are
 * concatenated over the segments, outTexts receives the segment texts NUL-separated. */
int32_t whisperc_debug_token_timestamps( const char* modelPath, const float* pcm, uint64_t nSamples, int32_t nSegments,
	const int64_t* segTimes, const int32_t* segTokenCounts, const int32_t* ids, const int32_t* tids, const float* p, const float* pt,
	const float* ptsum, float tholdPt, float tholdPtsum, int32_t maxLen, int32_t segCap, int32_t tokCap, int32_t* outSegCount,
	int64_t* outSegTimes, int32_t* outSegTokenCounts, char* outTexts, uint32_t textCap, int64_t* outTokTimes, float* outVlen );
/* Vocabulary and tokenizer of a model file on their own (host only, no device). whisperc_debug_tokenize returns the token count
 * (or a negative HRESULT); whisperc_debug_token_string writes the token's text and, optionally, the special ids in the order
 * eot, sot, prev, solm, not, beg, translate, transcribe (S_FALSE when the id has no string). */
int32_t whisperc_debug_tokenize( const char* modelPath, const char* text, int32_t* out, int cap );
int32_t whisperc_debug_token_string( const char* modelPath, int32_t token, char* out, uint32_t outCap, int32_t* specials8 );
/* Process-wide choice between the two host loops the reference ships: 0 (default) = its CPU model's whisper_full
 * (Whisper/source/whisper.cpp:2765-3120: drops the past prompt when < 5 s remain, retries a failed window once without it),
 * 1 = its GPU model's ContextImpl::runFullImpl (Whisper/Whisper/ContextImpl.cpp:452-793: neither rule). */
int32_t whisperc_set_host_loop_rules( int mode );
/* Where eSamplingStrategy::BeamSearch ranks a step's candidates: 0 (default) = on the device, the whole step a captured graph and the host polling
 * `done` every 16 steps (wh_beam_window_*); 1 = on the host after every step (wh_beam_candidates / wh_reorder_self_cache: the round-4 decoder, kept as
 * the checker of the device's restatement -- the two must give the same transcript). Process-wide. */
int32_t whisperc_set_beam_ranking( int onHost );
#ifdef __cplusplus
}
#endif
#endif
