/* whisper_hip.h -- C ABI of libwhisper_hip.so, the MI355X (gfx950) compute path.
 *
 * This is the drop-in boundary below the reference's host code: everything the reference implements in
 * Whisper/Whisper/WhisperContext.{h,cpp} (encode/decode graphs), Whisper/ML/MlContext.{h,cpp} (one method per
 * tensor op, each a D3D11 compute-shader dispatch) and the .hlsl files of ComputeShaders/ is replaced by the entry points
 * below.  Plain pointers and sizes only; no C++ types, no torch types.  All device pointers are HIP device
 * pointers; `stream` is a hipStream_t passed as void* (0 = the null stream).  Every function returns 0 on
 * success or a negative wh_status; wh_last_error() gives the text.  Nothing here ever falls back to the CPU.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference tree).
 */
#ifndef WHISPER_HIP_H
#define WHISPER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WH_API __attribute__( ( visibility( "default" ) ) )

typedef enum wh_status
{
	WH_OK = 0,
	WH_E_INVALIDARG = -1,	/* E_INVALIDARG in the reference's HRESULT vocabulary */
	WH_E_OUTOFMEMORY = -2,
	WH_E_HIP = -3,			/* a HIP runtime call failed; see wh_last_error() */
	WH_E_NOT_READY = -4,	/* model not finalized / encode not run before decode */
	WH_E_NO_DEVICE = -5,
	WH_E_BOUNDS = -6,
	WH_E_TIMEOUT = -7		/* a collective's deadline passed: a rank did not arrive (wh_comm_create_timeout, wh_comm_set_timeout) */
} wh_status;

WH_API const char* wh_last_error( void );

/* ---- device (replaces Whisper/D3D/createDevice.cpp, listGPUs.cpp; Whisper/ML/Device.cpp:92-124) ---- */
WH_API int wh_device_count( void );
/* name: >= 256 bytes. Mirrors the adapter enumeration behind Whisper::listGPUs (Whisper/API/iContext.cl.h:62-70). */
WH_API int wh_device_info( int device, char* name, size_t nameCap, uint64_t* totalMemBytes, int* computeUnits );
WH_API int wh_device_set( int device );

/* ---- model (replaces Whisper/Whisper/ModelBuffers.{h,cpp}, WhisperModel.cpp:257-340 "loadGpu") ----
 * sModelParams of the reference (Whisper/Whisper/sModelParams.h:5-18), same field order as the ggml file. */
typedef struct wh_hparams
{
	int32_t n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
	int32_t n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels, f16;
} wh_hparams;

typedef struct wh_model wh_model;

/* Size in bytes of the packed device weight arena for these hparams (FP16 matrices, FP32 vectors, our layout). */
WH_API int64_t wh_model_arena_bytes( const wh_hparams* hp );
/* arenaDev == NULL: the library hipMallocs the arena.  Otherwise the caller owns device memory of at least
 * wh_model_arena_bytes() bytes (e.g. a torch tensor that was just filled by an RCCL broadcast); pass
 * alreadyFilled != 0 if it already holds a finalized arena image from another rank with the same hparams. */
WH_API int wh_model_create( const wh_hparams* hp, void* arenaDev, int alreadyFilled, wh_model** out );
WH_API void wh_model_destroy( wh_model* m );
/* Upload one tensor of the ggml file by its file name ("encoder.blocks.3.attn.query.weight" ...;
 * name map Whisper/Whisper/WhisperModel.cpp:63-162).  ne[] in ggml order (ne[0] contiguous), nDims 1..3,
 * isF16 = the file's ftype != 0.  `data` is HOST memory, copied synchronously.  Unknown names, wrong shapes and
 * duplicates are errors, like the reference loader (WhisperModel.cpp:292-297, 331-335). */
WH_API int wh_model_set_tensor( wh_model* m, const char* name, int nDims, const int32_t* ne, int isF16, const void* data );
/* Mel filterbank from the file header, [n_mel][n_fft] FP32 (WhisperModel.cpp:456-470). */
WH_API int wh_model_set_filters( wh_model* m, int nMel, int nFft, const float* data );
/* Verifies every expected tensor arrived exactly once (WhisperModel.cpp:331-335) and builds derived data. */
WH_API int wh_model_finalize( wh_model* m );
WH_API int wh_model_arena( wh_model* m, void** dev, int64_t* bytes );
WH_API int wh_model_hparams( const wh_model* m, wh_hparams* out );

/* ---- one process per GPU: the weights travel over xGMI (RCCL), not through every rank's file system ----
 * The reference has one GPU per model and shares a loaded model between contexts of ONE device (iModel::clone,
 * Whisper/Whisper/ModelImpl.cpp:40-60; the adapter is chosen once, API/sModelSetup.h:30-34). Windows of a recording
 * (and recordings) are independent, so N GPUs hold N copies of the arena and split the windows; the only exchange is
 * this broadcast before the first window. The communicator is RCCL's (librccl.so is opened on first use, the library
 * does not link it): rank `root` calls wh_comm_unique_id and ships the 128 bytes to the other ranks by whatever the
 * host program has (a file, a socket, MPI, an environment variable); every rank -- one process per GPU, its device
 * selected with wh_device_set -- then calls wh_comm_create with the same id. wh_model_broadcast sends the FINALIZED
 * arena of `root` into the arena of every other rank (models created with the same hparams and arenaDev == NULL or the
 * caller's own buffer) and marks those models finalized; it returns the measured seconds in *secondsOut when non-NULL. */
#define WH_COMM_ID_BYTES 128
typedef struct wh_comm wh_comm;
/* Packaging check without a GPU: the collective library opens under one of the names the entry points below try (librccl.so.1, librccl.so, the same two
 * under /opt/rocm/lib) and exports every function they call; detail (optional) receives what was opened or what is missing. 0 or WH_E_NOT_READY. */
WH_API int wh_comm_runtime_check( char* detail, size_t detailCap );
WH_API int wh_comm_unique_id( void* id128 );
WH_API int wh_comm_create( const void* id128, int rank, int worldSize, wh_comm** out );
WH_API int wh_comm_destroy( wh_comm* comm );
WH_API int wh_comm_info( const wh_comm* comm, int* rank, int* worldSize );
WH_API int wh_model_broadcast( wh_model* m, wh_comm* comm, int root, double* secondsOut );
/* All ranks: blocks until every rank has arrived (a 4-byte all-reduce on the communicator's stream). */
WH_API int wh_comm_barrier( wh_comm* comm );
/* RCCL has no deadlines: a rank that died leaves its siblings inside ncclCommInitRank / a collective for ever. With a timeout the
 * rendezvous (wh_comm_create_timeout) and every later collective of the communicator (wh_comm_set_timeout; barrier, broadcasts) return
 * WH_E_TIMEOUT after `seconds` instead; the caller is expected to give the job up (whisper-mgpu exits, its parent ends the other ranks).
 * 0 = wait for ever (wh_comm_create). */
WH_API int wh_comm_create_timeout( const void* id128, int rank, int worldSize, double timeoutSeconds, wh_comm** out );
WH_API int wh_comm_set_timeout( wh_comm* comm, double seconds );
/* Four bytes from `root` to every rank: how a root that failed BEFORE a collective (a model file it could not read) tells the ranks
 * that are about to enter it -- failure has to be collective too. */
WH_API int wh_comm_broadcast_i32( wh_comm* comm, int root, int32_t* value );

/* ---- context (replaces DirectCompute::WhisperContext, Whisper/Whisper/WhisperContext.h:20-140) ----
 * One context owns activations, the FP16 self- and cross-attention KV caches (KeyValueBuffers.h:7-53) for up to
 * maxBatch independent 30 s windows that are processed in lock step, and its workspace.  Single-threaded use,
 * like the reference (Whisper/ML/Device.cpp:163-177); several contexts of one model may be driven concurrently (each
 * has its own stream): a decode step is a chain of small dependent launches, so the work of other contexts runs
 * underneath it.  Sizing: a decode step costs about the same for 1 and for 32 windows (the weight-streaming kernel
 * holds up to 32 rows), beyond 32 the decoder falls back to the tiled GEMM; 28 windows (four 198 s clips) per context
 * and three contexts in flight is what bench.py measures. */
typedef struct wh_context wh_context;

typedef enum wh_flags
{
	WH_FLAG_NONE = 0,
	/* Emulate the reference CPU path's FP16, thread-partitioned accumulation of the decoder P.V product
	 * (Whisper/source/ggml.c:4689-4735, 4615-4644) with `parityThreads` virtual threads, and feed the encoder's P.V product
	 * the reference's operand fp16( e / sum ) (ggml.c:6035-6046) instead of the exact FP16 e with the division applied to the
	 * FP32 result. Slow; for parity runs. */
	WH_FLAG_PARITY_PV = 1,
	/* Launch the greedy loop's kernels one by one instead of replaying the captured hipGraph (debugging aid). */
	WH_FLAG_NO_GRAPH = 2,
	/* Keep copies of the intermediates at the reference's Tracing probe points (Whisper/Whisper/WhisperContext.cpp:142-638,
	 * source/whisper.cpp:1121-1869) for wh_debug_read: "enc.temp1", "enc.layer0.in", "enc-KQV", "dec-KQV", "dec-KQV#2". */
	WH_FLAG_DEBUG_CAPTURE = 4,
	/* wh_encode and wh_decode compute the reference CPU path's arithmetic IN THE REFERENCE'S SUMMATION ORDER: ggml_vec_dot_f16's 32 chains and
	 * reduction tree for every product (Whisper/source/ggml.c:751-790), sequential double sums in LayerNorm (:4098-4156), the 65536-entry GELU /
	 * exp tables, flash_attn_f16's FP16 P (:5912-6097), and the decoder's FP16 key-by-key P.V over `parityThreads` thread ranges (:4689-4735).
	 * Cross-attention caches, logits and probabilities are then the reference's BITS at that thread count (tests/test_gpu_exact.py), which is what
	 * north_star's "within 1e-3 on logits" is checked against without the reference's own thread-count band (0.05 .. 0.5 on logits) in the way.
	 * One thread per output on the VALU, no MFMA: ~100 x slower than the timed kernels, never measured. Host-stepped decoding only (wh_decode +
	 * wh_sample_best; the device-side greedy loop and beam search refuse the flag). The caches it fills are the product's own, so a timed decode
	 * step can be run on top of exact caches and vice versa. */
	WH_FLAG_PARITY_EXACT = 8
} wh_flags;

WH_API int wh_context_create( wh_model* m, int maxBatch, void* stream, wh_context** out );
/* The same with `hypotheses` decoder sequences per window (1, 2, 3, 4, 5 or 8): the sequences b*hypotheses .. b*hypotheses +
 * hypotheses-1 decode window b and share ONE pass over its cross-attention K/V per step (each has its own self-attention
 * cache). The reference declares beam search without implementing it (Whisper/API/sFullParams.h:12-13, `beam_width`,
 * `n_best`): this is the data path a beam / best-of-n decoder needs, an extension with no reference oracle beyond
 * "every hypothesis computes what a lone sequence fed the same tokens computes". In every decode entry point `batch`
 * then counts SEQUENCES (a multiple of `hypotheses`); wh_encode's `batch` keeps counting windows. */
WH_API int wh_context_create_hyp( wh_model* m, int maxBatch, int hypotheses, void* stream, wh_context** out );
WH_API void wh_context_destroy( wh_context* c );
/* Binds the model's device to the calling thread (every wh_* call on a model or context does it as well): the
 * counterpart of Device::setForCurrentThread (Whisper/ML/Device.cpp:163-177). Call it before wh_buffer_alloc /
 * wh_buffer_free, which take no context. */
WH_API int wh_context_bind( wh_context* c );
WH_API int wh_context_set_flags( wh_context* c, uint32_t flags, int parityThreads );
/* sFullParams::audio_ctx (Whisper/API/sFullParams.h; ContextImpl.cpp:24, 55, 488-489 = whisper.cpp's exp_n_audio_ctx): the encoder runs on the first
 * `audioCtx` positions of a window -- 2 * audioCtx spectrogram frames from its offset, the first audioCtx rows of the positional embedding -- and the
 * decoder's cross-attention sees that many keys. 0 = the model's n_audio_ctx. Voids the context's encoder output and captured graphs; call it between
 * windows, not between wh_encode and the decode steps that belong to it. */
WH_API int wh_context_set_audio_ctx( wh_context* c, int audioCtx );
/* Blocks until everything queued on the context's stream has finished. When `stream` was NULL at creation the context
 * owns a non-blocking stream (the legacy null stream cannot be captured into a hipGraph), so callers that produce
 * inputs or consume device outputs on another stream must order the two themselves; this is the simple way. */
WH_API int wh_context_synchronize( wh_context* c );
/* {RAM, VRAM} accounting like getMemoryUse() in the reference (WhisperContext.cpp:641-666) */
WH_API int wh_context_memory( const wh_context* c, int64_t* vramBytes );

/* Plain device buffers for host code that does not include HIP headers (PCM in, spectrogram scratch). Replaces the
 * buffer helpers of Whisper/D3D/createBuffer.cpp. upload / download are synchronous on the context's stream. */
WH_API int wh_buffer_alloc( int64_t bytes, void** dev );
WH_API int wh_buffer_free( void* dev );
WH_API int wh_buffer_upload( wh_context* c, void* dev, const void* host, int64_t bytes );
WH_API int wh_buffer_download( wh_context* c, void* host, const void* dev, int64_t bytes );
/* Enqueues the copy on the context's stream and returns; `host` must stay valid (and should be pinned) until the stream
 * passes it. This is how a caller puts the PCM upload inside the pipeline instead of in front of it. */
WH_API int wh_buffer_upload_async( wh_context* c, void* dev, const void* host, int64_t bytes );

/* PCM -> log-mel on the GPU. Replaces Spectrogram::pcmToMel (Whisper/Whisper/Spectrogram.cpp:64-122) ==
 * log_mel_spectrogram (Whisper/source/whisper.cpp:2060-2180): hop 160, Hann 400, |DFT|^2 with the reference's
 * p[j]+=p[400-j] fold, 80x201 filterbank, log10 clamp, (global max - 8) clamp, (x+4)/4.
 * pcmDev: FP32 [nSamples] device; melDev: FP32 [n_mel][nLen] device with nLen = nSamples/160 (row = mel bin).
 * The normalisation maximum is over the whole buffer, as in runFull. */
WH_API int wh_mel_spectrogram( wh_context* c, const float* pcmDev, int64_t nSamples, float* melDev, int64_t* nLenOut );

/* The same for `batch` independent buffers of nSamples each (buffer b at pcmDev + b * pcmStride -> melDev + b * melStride, each normalised by its OWN
 * maximum: what `batch` calls of wh_mel_spectrogram give, bit for bit) in three launches instead of 3 x batch -- the independent 30 s chunks of the
 * batch path (a chunk alone is 188 workgroups: less than the chip). Strides in elements; melStride >= n_mel * (nSamples / 160). */
WH_API int wh_mel_spectrogram_batch( wh_context* c, const float* pcmDev, int64_t nSamples, int64_t pcmStride, int batch, float* melDev, int64_t melStride );

/* One window of a STREAMED spectrogram. Replaces MelStreamer::makeBuffer + makeTransposedBuffer
 * (Whisper/Whisper/MelStreamer.cpp:189-245, :125-187), what iContext::runStreamed feeds the encoder with: frames
 * [frame0, frame0 + nFrames) of the stream (frame f = 400 samples from f*160, zero past nSamples; frames >= nChunks, the
 * number of 160-sample chunks the reader delivered, are 0 BEFORE normalisation), clamped to (the WINDOW's maximum - 8) with
 * the maximum floored at 1e-20, then (x+4)/4 in FP32. reusePreviousMax != 0 normalises with the maximum the previous
 * call stored instead (the streamer does that when a shorter request ends at the same frame as the last one, :158-166).
 * melDev: FP32 [n_mel][nFrames]. */
WH_API int wh_mel_spectrogram_window( wh_context* c, const float* pcmDev, int64_t nSamples, int64_t frame0, int64_t nFrames, int64_t nChunks,
	int reusePreviousMax, float* melDev );

/* Encoder. Replaces WhisperContext::encode (WhisperContext.cpp:310-399) == whisper_encode (whisper.cpp:1084-1496).
 * melDev: FP32 device, `batch` spectrograms each [n_mel][melLen] (melStride floats apart); for each the window
 * [melOffset, melOffset + 2*n_audio_ctx) is taken and zero-padded (MelInputTensor.cpp:8-63).  Fills the
 * cross-attention caches of all decoder layers for batch slots 0..batch-1 and resets their self-attention state. */
WH_API int wh_encode( wh_context* c, const float* melDev, int batch, int64_t melLen, int64_t melStride, const int32_t* melOffsets );

/* The same for windows that come from DIFFERENT spectrograms (recordings of different lengths: the streams of Whisper::runFullBatch):
 * window b = frames [offset, offset + 2*n_audio_ctx) of the FP32 [n_mel][melLen] spectrogram at melDev, zero beyond its end
 * (MelInputTensor.cpp:8-63); melDev == NULL is a window of zeros (an idle slot of a lock-step batch). windows: HOST [batch]. */
typedef struct wh_mel_window
{
	const float* melDev;
	int64_t melLen;
	int32_t offset, reserved;
} wh_mel_window;
WH_API int wh_encode_windows( wh_context* c, const wh_mel_window* windows, int batch );

/* Decoder step. Replaces WhisperContext::decode (WhisperContext.cpp:578-639) == whisper_decode (whisper.cpp:1508-1872).
 * tokens: HOST int32 [batch][nTokens]; every sequence advances from position nPast by nTokens.
 * Outputs for the LAST token of each sequence (the only row the reference ever consumes, ContextImpl.cpp:159-169):
 *   logitsHost / probsHost: HOST FP32 [batch][n_vocab], either may be NULL (skips that download).
 * Synchronises the stream before returning when any host output is requested. */
WH_API int wh_decode( wh_context* c, const int32_t* tokens, int batch, int nTokens, int nPast, float* logitsHost, float* probsHost );

/* sTokenData of the reference (Whisper/Whisper/sTokenData.h): the result of ContextImpl::sampleBest. */
typedef struct wh_token_data
{
	int32_t id, tid;
	float p, pt, ptsum;
} wh_token_data;

/* Greedy sampling on the device, from the probabilities of the last wh_decode call. Replaces
 * ContextImpl::sampleBest / sampleTimestamp (Whisper/Whisper/ContextImpl.cpp:71-169): timestamp-vs-text rule,
 * initial-timestamp <= 1.00 s cap, first of the top-4 that is not sot/solm/not. forceTimestamp / isInitial are
 * per call (same for all sequences). out: HOST [batch]. Exact ties resolve to the lower token id. */
WH_API int wh_sample_best( wh_context* c, int batch, int forceTimestamp, int isInitial, wh_token_data* out );

/* Beam search on hypothesis groups (extension: the reference declares eSamplingStrategy::BeamSearch / beam_search.beam_width,
 * Whisper/API/sFullParams.h:10-13, and implements only the greedy strategy). Data path of one step, for the sequences of a
 * wh_context_create_hyp context:   wh_decode( tokens, batch, 1, nPast, NULL, NULL )  ->  wh_beam_candidates  ->  the host ranks
 * parent score + log p over each window's candidates and keeps the best `hypotheses`  ->  wh_reorder_self_cache( parents )  ->  next wh_decode.
 *   wh_beam_candidates: the `width` (1 .. 8) best continuations of every sequence from the probabilities of the last wh_decode, under
 *     sampleBest's own rules (timestamp-vs-text sum rule, initial-timestamp cap, sot / solm / not skipped): candidate 0 IS wh_sample_best's
 *     token, so width 1 is the greedy decoder. out: HOST [batch][width].
 *   wh_reorder_self_cache: sequence j continues sequence parents[j] (same window): rows [0, rows) of the self-attention caches of all layers are
 *     copied parents[j] -> j (through a scratch copy, so any permutation is safe; parents[j] == j costs nothing). The cross-attention K/V of a
 *     window are shared by its hypotheses and never move. */
WH_API int wh_beam_candidates( wh_context* c, int batch, int width, int forceTimestamp, int isInitial, wh_token_data* out );
WH_API int wh_reorder_self_cache( wh_context* c, int batch, const int32_t* parents, int rows );

/* Beam search WITHOUT the host in the loop (round 5; configs[2] of BASELINE.json). The per-step ranking above -- pool of live hypotheses x candidates
 * by parent score + log p, the best `width` accepted, the window's stop rules applied to each (WindowScan of libWhisper.so's host loop =
 * Whisper/Whisper/ContextImpl.cpp:597-673), finished / live lists, "can a live hypothesis still win" -- runs as a kernel behind every decode step,
 * writes the parents the next step's cache reorder reads and the tokens its embedding reads, and the whole step (reorder -> decode -> softmax ->
 * candidates -> ranking) is ONE captured graph replayed per token. The host enqueues chunks of steps and polls `done`.
 *   wh_beam_window_start    windows x (hypotheses of the context) sequences; promptTokens HOST [windows][nPrompt] (every slot of a window starts from
 *                           it); rules HOST [windows]; the prompt step, the first ranking (sampleTimestamp( true ) rules) and nSteps ranked steps are
 *                           enqueued. The call WAITS for what the stream holds first (the window's rules and initial state are copied synchronously, and the step
 *                           graph is captured on first use): one host round trip per window, the encoder included; the steps themselves never block. The
 *                           captured graph is keyed on (sequences, width): kernel options set with wh_debug_set_option / _tuning afterwards do not reach it.
 *                           forced != 0 in a window's rules: no stop rules, every hypothesis lives (random-weight workloads).
 *   wh_beam_window_continue nSteps more (bounded by n_text_ctx like wh_decode_window_continue). Steps enqueued after a window is done change nothing.
 *   wh_beam_window_status   blocks until everything enqueued has run; the search state of every window: HOST [windows].
 *   wh_beam_window_records  the accepted proposals of ranking steps [firstStep, firstStep + count): HOST [count][windows][width]; a hypothesis' token
 *                           chain is followed from its `rec` (= step * width + index) through `parent` down to -1 (parent == -2: unused entry).
 * Width 1 is the greedy decoder token for token. */
typedef struct wh_beam_rules
{
	int32_t seek, seekEnd;		/* the window's first frame and the stream's end, 10 ms units */
	int32_t nMax;				/* n_text_ctx / 2 - 4 */
	int32_t maxTokens;			/* sFullParams::max_tokens (0 = no limit) */
	int32_t singleSegment;
	int32_t tokenBeg, tokenEot;
	int32_t forced;
} wh_beam_rules;
typedef struct wh_beam_hyp
{
	double sum;					/* cumulative log-probability */
	int32_t i, hasTs, seekDelta, resultLen, failed, over;	/* WindowScan's state */
	int32_t nTok, rec;
} wh_beam_hyp;
typedef struct wh_beam_window
{
	int32_t step, nLive, nFinished, done;
	int32_t nPrompt, nTextCtx, reserved0, reserved1;
	wh_beam_hyp live[ 8 ];
	wh_beam_hyp finished[ 24 ];
} wh_beam_window;
typedef struct wh_beam_record
{
	wh_token_data t;
	int32_t parent, finished, reserved;
} wh_beam_record;
WH_API int wh_beam_window_start( wh_context* c, int windows, const int32_t* promptTokens, int nPrompt, int width, const wh_beam_rules* rules, int nSteps );
WH_API int wh_beam_window_continue( wh_context* c, int nSteps );
WH_API int wh_beam_window_status( wh_context* c, wh_beam_window* out );
WH_API int wh_beam_window_records( wh_context* c, int firstStep, int count, wh_beam_record* out );

/* The greedy loop of ContextImpl::runFullImpl (Whisper/Whisper/ContextImpl.cpp:597-673: decode -> sampleBest ->
 * feed the token back) kept on the device for nSteps tokens: firstTokens (HOST, [batch]) are fed at position nPast, each
 * step runs the decoder for one token per sequence, samples with the sampleBest rules and feeds the choice back, with
 * no host round trip in between (one captured hipGraph is replayed per token; position and flags live in device
 * memory). forceFirstTimestamp / firstIsInitial apply to the first sample only (sampleTimestamp(true) of the
 * reference). out: HOST [nSteps][batch]. The host applies its stop rules to the returned tokens afterwards; tokens
 * sampled after a sequence's stop condition are simply discarded by the caller. */
WH_API int wh_decode_greedy( wh_context* c, int batch, const int32_t* firstTokens, int nPast, int nSteps, int forceFirstTimestamp,
	int firstIsInitial, wh_token_data* out );

/* One whole window's decode, enqueued without blocking the host: prompt step (HOST tokens [batch][nPrompt] at position 0)
 * -> first sample -> nSteps greedy steps; wh_decode_window_finish blocks and returns the 1 + nSteps samples as HOST
 * [1 + nSteps][batch]. Contexts own their streams, so several windows (or groups of windows) started back to back from
 * one host thread overlap on the GPU: a single-token decode step is latency-bound and occupies a fraction of the CUs. */
WH_API int wh_decode_window_start( wh_context* c, int batch, const int32_t* promptTokens, int nPrompt, int nSteps, int forceFirstTimestamp,
	int firstIsInitial );
WH_API int wh_decode_window_finish( wh_context* c, wh_token_data* out );
/* The same for a lock-step batch whose sequences carry prompts of DIFFERENT lengths -- the streams of a batch scheduler (Whisper::runFullBatch
 * of libWhisper.so): every stream keeps the reference's sequential semantics, so its window's prompt is [prev] + its own past text + the task
 * tokens (ContextImpl.cpp:565-576) and the streams of one batch stand at different decoder positions. promptTokens: HOST [batch][nPromptMax],
 * row b = promptLens[b] tokens followed by padding (ignored); 1 <= promptLens[b] <= nPromptMax. Sequence b then computes exactly what it computes
 * alone: its tokens sit at positions 0 .. promptLens[b]-1, its greedy step s feeds position promptLens[b] + s, its self-attention sees its own
 * keys only (positions live per sequence in device memory; the padded rows of a shorter prompt are computed and never consumed: their cache
 * rows are overwritten by the sequence's own later tokens before any query can see them). One hypothesis per window.
 * nPromptMax + nSteps (and every later wh_decode_window_continue) is bounded by n_text_ctx. */
WH_API int wh_decode_window_start_ragged( wh_context* c, int batch, const int32_t* promptTokens, const int32_t* promptLens, int nPromptMax, int nSteps,
	int forceFirstTimestamp, int firstIsInitial );
/* nSteps more greedy steps of the window in progress (no host round trip: position, last token and sampler flags are in
 * device memory), and a blocking read of samples [first, first + count) -> HOST [count][batch] that waits for those
 * samples only. The reference's loop looks at every token before it decodes the next one (ContextImpl.cpp:597-673); a
 * caller that keeps one chunk queued behind the one it is scanning loses at most that chunk when a stop token shows up. */
WH_API int wh_decode_window_continue( wh_context* c, int nSteps );
WH_API int wh_decode_window_fetch( wh_context* c, int first, int count, wh_token_data* out );
/* Non-blocking: 1 when samples [first, first + count) of the window in progress exist (wh_decode_window_fetch would not wait), 0 when not
 * yet, negative on error. Lets ONE host thread serve several contexts (a scheduler that keeps two lock-step batches in flight). */
WH_API int wh_decode_window_ready( wh_context* c, int first, int count );

/* Per-kernel-class GPU timings, the counterpart of the reference's GpuProfiler / iContext::timingsPrint
 * (Whisper/Utils/GpuProfiler.h:21-188, Whisper/Whisper/ContextImpl.misc.cpp:170-182). hipEvent pairs around every launch
 * on the context's stream while enabled; flops / bytes are the algorithmic work of the launches. */
typedef struct wh_profile_entry
{
	char name[ 32 ];
	int64_t calls;
	double ms, flops, bytes;
} wh_profile_entry;
WH_API int wh_profile_enable( wh_context* c, int on );	/* resets the counters */
WH_API int wh_profile_read( wh_context* c, wh_profile_entry* out, int cap, int* count );

/* Test / parity access to internal state (the reference reaches these through its Tracing probe points,
 * Whisper/Whisper/WhisperContext.cpp:142-638). All outputs HOST FP32.
 *   what = "encode-out"  [batch][n_ctx][d]           (only valid right after wh_encode)
 *          "exp-table"   [0x5000]: the model's copy of the reference's table_exp_f16 (ggml.c:1375-1385), entry i = fp16( expf( -|fp16 bits i| ) )
 *          "cross-k" / "cross-v"   layer, [batch][n_ctx][d] token-major like the reference's kvCross
 *          "self-k" / "self-v"     layer, [sequences][rows][d]
 * and, captured under WH_FLAG_DEBUG_CAPTURE by the wh_encode / wh_decode call that follows the flag:
 *          "enc.temp1"      conv1 + bias + GELU, [batch][2*n_ctx][d] (time-major; the reference's tensor is [d][2*n_ctx])
 *          "enc.layer0.in"  conv2 + GELU + positional embedding = input of encoder layer 0, [batch][n_ctx][d]
 *          "enc-KQV"        encoder layer 0 attention output, [batch][n_ctx][d] (heads side by side)
 *          "dec-KQV" / "dec-KQV#2"   decoder layer 0 self / cross attention output of the last wh_decode, [rows][d]
 */
WH_API int wh_debug_read( wh_context* c, const char* what, int layer, int rows, float* dstHost, int64_t dstCapFloats );

/* Development micro-benchmarks (tools/gemm_probe.py): kind 0 = chain of empty kernels (variant = workgroups), 2 = the same
 * replayed from a hipGraph, 1 = tiled GEMM tile-shape variant on an M x N x K problem. Returns milliseconds per iteration. */
/* Bit mask of kernel-variant switches (whisper_amd/csrc/kernels.h eTuning) for in-process A/B runs; contexts created
 * afterwards (and their captured graphs) use the new setting. */
WH_API int wh_debug_set_tuning( uint32_t mask );
/* Integer knobs beyond the 32 switches (whisper_amd/csrc/kernels.h struct Options: "dec_tile", "dec_depth", "dec_wide_rows", "dec_deep_rows", "vocab_decrows",
 * "enc_chunk", "self_fuse_max_rows", "self_nq", "self_wave_min_rows", "enc_exp", "exact_enc_layers", "exact_alt_order", "gemm_mf16", "dec_lds", "dec_lds_ks", "dec_split", "cross_mfma", "vocab_lds", "beam_regs", "reorder_group", "gemm_big_min_rows"); also settable as WH_OPT_<NAME> in the environment at load.
 * Unknown names and values outside [-1, 2^24]: WH_E_INVALIDARG (the environment form: ignored with a line on stderr). The options are process-global and read without
 * synchronisation by every launch: set them BEFORE contexts are created, never while another thread runs one. */
WH_API int wh_debug_set_option( const char* name, int value );
/* The current value of an option (its compiled-in default until wh_debug_set_option or WH_OPT_<NAME> changes it). Host only: works without a device. */
WH_API int wh_debug_get_option( const char* name, int* value );
WH_API int wh_debug_probe( wh_context* c, int kind, int variant, int M, int N, int K, int iters, float* msPerIter );

/* ---- op-level entry points (replace the MlContext methods, Whisper/ML/MlContext.h:13-113). Device pointers. ---- */

/* MlContext::mulMat with an FP16 weight (ComputeShaders/mulMatTiled.hlsl, mulMatByRowTiled.hlsl):
 * out[m][n] = sum_k fp16(a[m][k]) * w[n][k] (+ bias[n]) (+ residual[m][n]); a: FP16 [M][K] (already rounded, which is
 * what ggml does at ggml.c:4588-4611), w: FP16 [N][K], out FP32 [M][N]. bias/residual may be NULL. */
WH_API int wh_op_mul_mat( void* stream, const void* aF16, const void* wF16, const float* bias, const float* residual,
	float* out, int M, int N, int K );
/* mulMat + addRepeatGelu (ComputeShaders/addRepeatGelu.hlsl): out FP16 [M][N] = gelu16( acc + bias ) */
WH_API int wh_op_mul_mat_gelu( void* stream, const void* aF16, const void* wF16, const float* bias, void* outF16, int M, int N, int K );
/* MlContext::norm + fmaRepeat (norm.hlsl, fmaRepeat1.hlsl): out FP16 [rows][d] = fp16( norm(x) * w + b ) */
WH_API int wh_op_layer_norm( void* stream, const float* x, const float* w, const float* b, void* outF16, int rows, int d );
/* MlContext::flashAttention, unmasked (flashAttention.hlsl:76-169 == ggml.c:5912-6097).
 * q, k: FP16 [batch*heads][nCtx][64]. vFrag: FP16 [batch*heads][64 * nCtxPad], nCtxPad = roundup(nCtx, 256), zero beyond
 * nCtx, in the operand order of the P.V matrix instruction (what the QKV product's epilogue writes):
 *   index(key, dd) = (((key>>4)*2 + (dd>>5))*64 + ((key>>2)&1)*32 + (dd&31))*8 + ((key>>3)&1)*4 + (key&3)
 * out: FP16 [batch][nCtx][heads*64]. */
WH_API int wh_op_flash_attention( void* stream, const void* q, const void* k, const void* vFrag, void* out, int batch, int heads, int nCtx );
/* Decoder attention of WhisperContext::decodeLayer (Whisper/Whisper/WhisperContext.cpp:455-470, 505-519: mulMat(K,Q) ->
 * diagMaskInf -> softMax -> mulMat(V,.)) on caches laid out [block][head][keyStride][64] FP16. q, out: FP16
 * [sequences*nTok][heads*64], q already scaled. causal: query i of a sequence sees keys <= nPast + i. `group` consecutive
 * sequences share one cache block (cross-attention of a window's hypotheses); parityThreads > 0 emulates the CPU path's
 * FP16 thread-partitioned P.V (ggml.c:4689-4735). */
WH_API int wh_op_decoder_attention( void* stream, const void* qF16, const void* kCache, const void* vCache, void* outF16, int sequences, int heads,
	int nTok, int nKeys, int keyStride, int causal, int nPast, int group, int parityThreads );
/* The cross-attention half of a decode step in one launch (WhisperContext.cpp:489-519): q = fp16( ( Wq . fp16( LayerNorm(x)
 * * lnW + lnB ) + qB ) * qScale ) per head inside the attention kernel; x: FP32 [sequences][heads*64]. */
WH_API int wh_op_decoder_cross_attention( void* stream, const float* x, const float* lnW, const float* lnB, const void* qW, const float* qB,
	float qScale, const void* kCache, const void* vCache, void* outF16, int sequences, int heads, int nKeys, int keyStride, int group );
/* softMax over rows with the reference's FP16 exp table semantics (softMax.hlsl / ggml.c:5030-5090): in place, FP32 */
WH_API int wh_op_soft_max( void* stream, float* x, int rows, int cols );

#ifdef __cplusplus
}
#endif
#endif
