// Model arena, context and the encode / decode graphs behind the C ABI of include/whisper_hip.h.
//
// This file is the MI355X counterpart of the reference's DirectCompute::WhisperContext
// (Whisper/Whisper/WhisperContext.cpp: encode :310-399, encodeLayer :158-289, decode :578-639, decodeLayer :407-576),
// ModelBuffers (Whisper/Whisper/ModelBuffers.h:8-112) and KeyValueBuffers (KeyValueBuffers.h:7-53). It is host code
// only: every arithmetic step is a kernel from gemm.hip / attn_enc.hip / attn_dec.hip / elementwise.hip / mel.hip.
//
// Memory model (sized for 288 GB of HBM3E, no allocation in steady state):
//   * ONE packed weight arena per model, layout a pure function of the hparams, so a rank that did not read the file
//     can receive it with a single RCCL broadcast. Q/K/V weights of a layer are concatenated to one [3d][d] matrix, the
//     cross-attention K/V weights of ALL decoder layers to one [2*L*d][d] matrix (one big GEMM per window).
//   * per context: activations for maxBatch windows in lock step + FP16 KV caches
//       cross  [layer][batch][head][n_audio_ctx][64]   (K pre-scaled by (d/H)^-0.25, whisper.cpp:1465)
//       self   [layer][batch][head][n_text_ctx][64]
#include "kernels.h"
#include "../../include/whisper_hip.h"
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <condition_variable>
#include <memory>
#include <cstdlib>
#include <set>
#include <string>
#include <vector>
#include <chrono>
#include <dlfcn.h>
#include <atomic>
#include <thread>

// ncclUniqueId by value, as ncclCommInitRank takes it (rccl.h: struct { char internal[128]; })
struct ncclUniqueIdBlob { char internal[ 128 ]; };

namespace wh
{
	unsigned g_tuning = TUNE_DEFAULT;
	Options g_opt;
	namespace
	{
		struct OptionName { const char* name; int Options::* field; };
		const OptionName g_optionNames[] = { { "dec_tile", &Options::decTile }, { "dec_depth", &Options::decDepth }, { "dec_wide_rows", &Options::decWideRows }, { "dec_deep_rows", &Options::decDeepRows }, { "vocab_decrows", &Options::vocabDecRows }, { "enc_chunk", &Options::encChunk },
			{ "self_fuse_max_rows", &Options::selfFuseMaxRows }, { "self_nq", &Options::selfNq }, { "self_wave_min_rows", &Options::selfWaveMinRows }, { "exact_enc_layers", &Options::exactEncLayers }, { "exact_alt_order", &Options::exactAltOrder }, { "enc_exp", &Options::encExp }, { "enc_ablate", &Options::encAblate }, { "gemm_mf16", &Options::gemmMf16 }, { "dec_lds", &Options::decLds }, { "dec_lds_ks", &Options::decLdsKs }, { "dec_split", &Options::decSplit }, { "vocab_lds", &Options::vocabLds }, { "beam_regs", &Options::beamRegs }, { "reorder_group", &Options::reorderGroup }, { "gemm_big_min_rows", &Options::gemmBigMinRows }, { "cross_mfma", &Options::crossMfma } };
		// WH_OPT_DEC_TILE=44 ... at load
		const bool g_optionsFromEnv = []()
		{
			for( const OptionName& o : g_optionNames )
			{
				std::string env = "WH_OPT_";
				for( const char* p = o.name; *p; p++ ) env.push_back( (char)toupper( (unsigned char)*p ) );
				if( const char* e = getenv( env.c_str() ) )
				{
					// a number in the options' common range, or the variable is ignored (garbage used to become 0 and re-route kernels silently)
					char* end = nullptr;
					const long v = strtol( e, &end, 10 );
					if( end != e && *end == 0 && v >= -1 && v <= ( 1 << 24 ) ) g_opt.*( o.field ) = (int)v;
					else fprintf( stderr, "[wh] %s='%s' ignored (not an integer in [-1, 2^24])\n", env.c_str(), e );
				}
			}
			return true;
		}();
	}
	static thread_local std::string g_lastError;
	void setError( const std::string& s ) { g_lastError = s; }
	int hipFail( hipError_t e, const char* what, const char* file, int line )
	{
		char buf[ 512 ];
		snprintf( buf, sizeof( buf ), "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString( e ), file, line, what );
		g_lastError = buf;
		return WH_E_HIP;
	}
}
using namespace wh;

namespace
{
	__global__ void probeEmpty( int* p ) { if( p && threadIdx.x == 0xFFFF ) *p = 1; }
	inline int64_t align256( int64_t x ) { return ( x + 255 ) & ~(int64_t)255; }
	inline int roundUp( int x, int m ) { return ( x + m - 1 ) / m * m; }
	// conv1 as an implicit GEMM: K = 3 taps * n_mels channels, zero-padded to a multiple of 64 (80 mels: 240 -> 256;
	// the 128 mels of the large-v3 shape: 384 exactly)
	constexpr int CONV1_KPAD_MAX = 512;
	constexpr int MEL_BATCH_MAX = 1024;	   // buffers per launch of wh_mel_spectrogram_batch (a maximum each in the context's scratch)
	inline int conv1Kpad( const wh_hparams& hp ) { return roundUp( 3 * hp.n_mels, 64 ); }

	// Special token ids follow from the vocabulary size (Whisper/Whisper/Vocabulary.h:27-41 hard-codes 51864 / 51865):
	// every language token added to the multilingual vocabulary moves the ids behind the language block up by one.
	// 51864 (.en) -> extra 0, 51865 (multilingual, 99 languages) -> 1, 51866 (the large-v3 shape, 100 languages) -> 2.
	struct SpecialIds { int sot, solm, tnot, beg; };
	inline SpecialIds specialIds( const wh_hparams& hp )
	{
		const int extra = hp.n_vocab > 51864 ? hp.n_vocab - 51864 : 0;
		return SpecialIds{ 50257 + ( extra > 0 ? 1 : 0 ), 50361 + extra, 50362 + extra, 50363 + extra };
	}

	struct EncLayer
	{
		int64_t ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, w1, b1, w2, b2;
	};
	struct DecLayer
	{
		int64_t ln1w, ln1b, wqkv, bqkv, wo, bo, lncw, lncb, wcq, bcq, wco, bco, ln2w, ln2b, w1, b1, w2, b2;
	};
	struct Layout
	{
		int64_t filters, dft, expTab, encPe, conv1w, conv1b, conv2w, conv2b, lnPostW, lnPostB;
		int64_t decPe, te, decLnW, decLnB, wcross, bcross;
		std::vector<EncLayer> enc;
		std::vector<DecLayer> dec;
		int64_t total;
	};

	Layout makeLayout( const wh_hparams& hp )
	{
		Layout L;
		int64_t o = 0;
		auto take = [ & ]( int64_t bytes ) { const int64_t r = o; o = align256( o + bytes ); return r; };
		const int64_t d = hp.n_audio_state, V = hp.n_vocab;
		L.filters = take( 4ll * hp.n_mels * 201 );
		L.dft = take( 8ll * 800 );
		L.expTab = take( 2ll * EXP_TABLE_ENTRIES );
		L.encPe = take( 4ll * hp.n_audio_ctx * d );
		L.conv1w = take( 2ll * d * conv1Kpad( hp ) );
		L.conv1b = take( 4 * d );
		L.conv2w = take( 2ll * d * 3 * d );
		L.conv2b = take( 4 * d );
		L.lnPostW = take( 4 * d );
		L.lnPostB = take( 4 * d );
		L.enc.resize( hp.n_audio_layer );
		for( auto& e : L.enc )
		{
			e.ln1w = take( 4 * d ); e.ln1b = take( 4 * d );
			e.wqkv = take( 2ll * 3 * d * d ); e.bqkv = take( 4 * 3 * d );
			e.wo = take( 2ll * d * d ); e.bo = take( 4 * d );
			e.ln2w = take( 4 * d ); e.ln2b = take( 4 * d );
			e.w1 = take( 2ll * 4 * d * d ); e.b1 = take( 4 * 4 * d );
			e.w2 = take( 2ll * 4 * d * d ); e.b2 = take( 4 * d );
		}
		L.decPe = take( 4ll * hp.n_text_ctx * d );
		L.te = take( 2ll * V * d );
		L.decLnW = take( 4 * d ); L.decLnB = take( 4 * d );
		L.wcross = take( 2ll * 2 * hp.n_text_layer * d * d );
		L.bcross = take( 4ll * 2 * hp.n_text_layer * d );
		L.dec.resize( hp.n_text_layer );
		for( auto& e : L.dec )
		{
			e.ln1w = take( 4 * d ); e.ln1b = take( 4 * d );
			e.wqkv = take( 2ll * 3 * d * d ); e.bqkv = take( 4 * 3 * d );
			e.wo = take( 2ll * d * d ); e.bo = take( 4 * d );
			e.lncw = take( 4 * d ); e.lncb = take( 4 * d );
			e.wcq = take( 2ll * d * d ); e.bcq = take( 4 * d );
			e.wco = take( 2ll * d * d ); e.bco = take( 4 * d );
			e.ln2w = take( 4 * d ); e.ln2b = take( 4 * d );
			e.w1 = take( 2ll * 4 * d * d ); e.b1 = take( 4 * 4 * d );
			e.w2 = take( 2ll * 4 * d * d ); e.b2 = take( 4 * d );
		}
		L.total = o;
		return L;
	}

	int checkHparams( const wh_hparams* hp )
	{
		if( !hp ) { setError( "hparams is null" ); return WH_E_INVALIDARG; }
		const int d = hp->n_audio_state;
		if( d <= 0 || d != hp->n_text_state || ( d % 64 ) != 0 || hp->n_audio_head * HEAD_DIM != d || hp->n_text_head * HEAD_DIM != d )
		{
			setError( "unsupported model: need n_audio_state == n_text_state == 64 * heads" );
			return WH_E_INVALIDARG;
		}
		if( hp->n_audio_ctx <= 0 || hp->n_audio_ctx > 1536 || hp->n_text_ctx <= 0 || hp->n_text_ctx > 1536 || hp->n_mels <= 0 ||
			3 * hp->n_mels > CONV1_KPAD_MAX || ( hp->n_mels % 8 ) != 0 || hp->n_vocab <= 0 || hp->n_audio_layer <= 0 || hp->n_text_layer <= 0 )
		{
			setError( "unsupported model dimensions" );
			return WH_E_INVALIDARG;
		}
		return 0;
	}
}	// namespace

struct wh_model
{
	wh_hparams hp;
	Layout L;
	uint8_t* arena = nullptr;
	bool ownsArena = false;
	bool finalized = false;
	std::set<std::string> loaded;
	bool filtersSet = false;
	int device = 0;	   // the HIP device the arena lives on; every entry point binds the calling thread to it
	template<class T> T* at( int64_t off ) const { return (T*)( arena + off ); }
	size_t expectedTensors() const { return 11 + 15 * (size_t)hp.n_audio_layer + 24 * (size_t)hp.n_text_layer; }
};

// HIP's current device is per host thread. The reference binds the model's device to the calling thread at the top of
// every call (Device::setForCurrentThread, Whisper/ML/Device.cpp:163-177); so do we, which is what lets a context be
// created or run from a thread other than the one that loaded the model (iModel::clone, sModelSetup.adapter).
static int bindDevice( const wh_model* m )
{
	int cur = -1;
	if( hipGetDevice( &cur ) == hipSuccess && cur == m->device ) return 0;
	WH_HIP( hipSetDevice( m->device ) );
	return 0;
}
#define WH_BIND( model ) WH_CHECK( bindDevice( model ) )

// Per-kernel-class GPU timing, the counterpart of the reference's GpuProfiler (Whisper/Utils/GpuProfiler.h:21-188: a
// timestamp query per shader dispatch, aggregated per eComputeShader). hipEvent pairs on the context's stream; only
// active between wh_profile_enable(1) and wh_profile_read, because two event records per launch perturb launch-bound code.
enum eKernelClass : int
{
	KC_GEMM_TILED = 0, KC_GEMM_SKINNY, KC_GEMV, KC_ATTN_ENC, KC_ATTN_DEC, KC_ATTN_DEC_CROSS, KC_SELF_BLOCK, KC_LAYER_NORM, KC_MEL, KC_MEL_TO_CONV, KC_EMBED, KC_SOFTMAX,
	KC_SAMPLE, KC_EVENT_PAIR, KC_LAYER_NORM_DEC, KC_GEMM_DEC, KC_COUNT
};
// "attentionDecCross" = cross-attention launches (attentionDecG<NQ, true> / <NQ, false> with group or nKeys = n_audio_ctx),
// "attentionDec" = causal self-attention; "eventPair" = the calibration launches of wh_profile_enable (an empty kernel
// between the same two event records: what the bracket itself costs, to be subtracted from every per-launch average).
static const char* const kernelClassNames[ KC_COUNT ] = { "gemmTiled", "gemmSkinny", "gemvFused", "attentionEnc", "attentionDec", "attentionDecCross",
	"selfBlockDec", "layerNorm", "mel", "melToConvInput", "embed", "vocabSoftMax", "softMaxSample", "eventPair", "layerNormDec", "gemmDecode" };
// "layerNormDec" / "gemmDecode" = the LayerNorm launches and the M-tiled products of the DECODER graph (prompt steps; the vocabulary product of more than
// 128 sequences): kept apart from the encoder's, whose classes are the MFMA roofline of the bench line

struct Profiler
{
	bool on = false;
	struct Pending { int kc; hipEvent_t a, b; };
	std::vector<Pending> pending;
	std::vector<hipEvent_t> pool;
	int64_t calls[ KC_COUNT ] = {};
	double ms[ KC_COUNT ] = {}, flops[ KC_COUNT ] = {}, bytes[ KC_COUNT ] = {};
	hipEvent_t get()
	{
		if( !pool.empty() ) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
		hipEvent_t e = nullptr;
		(void)hipEventCreate( &e );
		return e;
	}
	void resolve()
	{
		for( const Pending& p : pending )
		{
			float t = 0;
			if( hipEventSynchronize( p.b ) == hipSuccess && hipEventElapsedTime( &t, p.a, p.b ) == hipSuccess ) ms[ p.kc ] += t;
			pool.push_back( p.a );
			pool.push_back( p.b );
		}
		pending.clear();
	}
	void reset()
	{
		resolve();
		for( int i = 0; i < KC_COUNT; i++ ) { calls[ i ] = 0; ms[ i ] = flops[ i ] = bytes[ i ] = 0; }
	}
	~Profiler()
	{
		resolve();
		for( hipEvent_t e : pool ) (void)hipEventDestroy( e );
	}
};

struct wh_context
{
	wh_model* m = nullptr;
	Profiler prof;
	int maxBatch = 0;	   // 30 s windows (encoder batch, cross-attention caches)
	int hyp = 1;		   // decoder hypotheses per window: rows b*hyp .. b*hyp+hyp-1 share window b's cross-attention K/V
	int maxSeq = 0;		   // decoder sequences = maxBatch * hyp (self-attention caches, logits, sampler state)
	hipStream_t stream = nullptr;
	uint32_t flags = 0;
	int parityThreads = 1;
	int T = 0, Tpad = 0, maxRows = 0;
	int64_t vram = 0;
	bool encoded = false;
	int lastBatch = 0;	   // decoder sequences of the last decode call
	int lastEncBatch = 0;  // windows of the last wh_encode
	int encChunk = 0;	   // windows the ENCODER runs at a time: its activations are sized for this many, a larger lock-step batch is encoded in
						   // equal chunks (the products are MFMA-bound and saturated at ~100 windows; only the cross-attention caches hold all windows)
	// encoder activations
	f16 *convIn = nullptr, *conv1Out = nullptr, *xn = nullptr, *q = nullptr, *k = nullptr, *vT = nullptr, *attn = nullptr, *h = nullptr;
	float* x = nullptr;
	int64_t convInStride = 0, conv1Stride = 0;
	// caches
	f16 *crossK = nullptr, *crossV = nullptr, *selfK = nullptr, *selfV = nullptr;
	// decoder activations
	float *dx = nullptr, *logits = nullptr, *probs = nullptr;
	f16 *dxn = nullptr, *dq = nullptr, *dattn = nullptr, *dh = nullptr;
	float* splitK = nullptr;	 // [8][min( maxRows, 128 )][d]: partial tiles of the K-split MLP down-projection (option dec_split)
	// single-stream decode steps (decode1.hip): cross-attention scores, per-split maxima and partial results of up to 4 sequences
	float *crossScores = nullptr, *crossSplitMax = nullptr, *crossPart = nullptr;
	int* tokensDev = nullptr;
	int* melOffsetsDev = nullptr;
	MelWindow* melWindowsDev = nullptr;
	TokenData* tokDataDev = nullptr;
	uint8_t* sampleScratch = nullptr;		   // TUNE_SAMPLE_SPREAD: slice records of the spread sampler (allocated on first use, before any capture)
	TokenData* beamCand = nullptr;			   // beam search: [maxSeq][8] candidates (allocated on first use)
	f16 *selfKScratch = nullptr, *selfVScratch = nullptr;	   // beam search: the copy a cache reorder goes through (allocated on first use)
	// beam search on the device (wh_beam_window_*): per-window rules and state, the records of every step, the parents a step's reorder reads
	BeamRules* beamRules = nullptr;
	BeamWindow* beamState = nullptr;
	BeamRecord* beamRecords = nullptr;
	int* beamParents = nullptr;
	hipGraphExec_t beamGraphExec = nullptr;
	int beamGraphBatch = 0, beamGraphWidth = 0;
	int beamWindows = 0, beamWidth = 0, beamSteps = 0;	   // the window in progress: windows, width, ranking steps enqueued so far
	float* melScratch = nullptr;
	// device-side greedy loop: the sampler's state and one position per sequence (the sequences of a lock-step batch may differ)
	DecodeState* state = nullptr;
	int* seqPos = nullptr;
	// host mailbox of the greedy loop (pinned, coherent): [n_text_ctx * maxSeq] records + stamps, and the generation of the window in progress
	TokenData* mailData = nullptr;
	int* mailFlag = nullptr;
	SampleMailbox mailDev = { nullptr, nullptr };
	int mailGen = 0;		   // generation of the window in progress (0 = its samples are not mirrored)
	int mailCounter = 0;	   // last generation handed out
	TokenData* greedyOut = nullptr;	   // [n_text_ctx][maxBatch]
	hipGraphExec_t graphExec = nullptr;
	int graphBatch = 0;
	uint32_t graphKey = 0;
	const int* raggedLastPos = nullptr;	   // set around the prompt step of a window whose prompts differ in length: device [batch], position of each sequence's last prompt token
	int windowSamples = 0;
	int windowPos = 0;		   // position the next greedy step feeds (prompt length + steps enqueued so far)
	hipStream_t copyStream = nullptr;
	// TUNE_SPLIT_STREAMS: the MFMA-bound encoder on its own low-priority stream, so that the latency-bound decode chains of
	// OTHER contexts (high-priority streams) get their workgroups dispatched first whenever CUs free up
	hipStream_t encStream = nullptr;
	hipEvent_t encReady = nullptr, encDone = nullptr;
	hipEvent_t encGateEv = nullptr;	   // TUNE_ENC_SERIAL: recorded behind this context's encoder; the next context's encoder waits for it
	int encCus = 0, totalCus = 0;	   // WH_ENC_CUS: CUs of the encoder stream's mask (0 = no spatial split)
	struct Mark { int endSample; hipEvent_t ev; };
	std::vector<Mark> marks;   // after each enqueued chunk of samples: an event wh_decode_window_fetch can wait for
	std::vector<hipEvent_t> markPool;
	// WH_FLAG_DEBUG_CAPTURE: copies of intermediates at the reference's Tracing probe points (WhisperContext.cpp:142-638)
	f16 *capTemp1 = nullptr, *capEncKqv = nullptr, *capDecKqvSelf = nullptr, *capDecKqvCross = nullptr;
	float* capLayer0In = nullptr;
	int capDecRows = 0;
	int profKeysHint = 1;	   // profiler only: keys a device-positioned self-attention launch sees (host mirror of the largest position + 1)
	// WH_FLAG_PARITY_EXACT (exact.hip): FP32 activations of the reference-order graph, allocated on first use for up to EXACT_CHUNK windows at a time
	// (a larger batch is encoded chunk by chunk), the decoder's rows and score scratch grown on demand; ggml_init's two 65536-entry tables
	struct Exact
	{
		static constexpr int CHUNK = 8;
		f16 *gelu = nullptr, *expt = nullptr;
		float *x = nullptr, *cur = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *kqv = nullptr, *h = nullptr, *conv1 = nullptr;
		float *dx = nullptr, *dcur = nullptr, *dq = nullptr, *dk = nullptr, *dv = nullptr, *dkqv = nullptr, *dh = nullptr, *scores = nullptr;
		int64_t decRows = 0, scoreFloats = 0;
		int encWindows = 0;
		std::vector<void*> owned;
	} ex;
	bool ownsStream = false;
	// pinned host staging for fully asynchronous enqueues: ints [0, 4096) window offsets or descriptors (6 ints each: up to 682 windows),
	// [4096, 4104) the sampler state, [4104, 4104 + maxSeq) positions, then the prompt tokens of a window (up to n_text_ctx per sequence)
	int32_t* pinned = nullptr;
	int64_t pinnedInts = 0;
	static constexpr int PIN_WINDOWS = 4096, PIN_STATE = PIN_WINDOWS, PIN_POS = PIN_STATE + 8;
	int32_t* pinTokens() const { return pinned + PIN_POS + maxSeq; }
	int64_t pinTokenCap() const { return pinnedInts - PIN_POS - maxSeq; }
	struct Allocation { void* base; void* body; int64_t bytes; const char* name; };
	std::vector<Allocation> allocations;

	// Every buffer starts zeroed. MUST_BE_ZERO marks the ones whose correctness depends on it: the convolution padding
	// rows, the V operand padding, the K/V caches (rows beyond n_past are masked, not skipped: they must be finite) and the
	// device-resident decode state. The others are written before they are read and are zeroed only as hygiene -- which is
	// exactly what WH_DEBUG_POISON=<byte> checks: in that mode those buffers are filled with the byte instead (0xFF = NaN in
	// FP16 / FP32 and -1 as an index), every buffer gets a guard region on both sides, and wh_context_destroy verifies the
	// guards. A run whose results or faults change under WH_DEBUG_POISON depends on stale device memory.
	enum eInit { DONT_CARE, MUST_BE_ZERO };
	template<class T> int alloc( T*& p, int64_t count, eInit init, const char* name )
	{
		void* v = nullptr;
		const int64_t bytes = count * (int64_t)sizeof( T );
		const int64_t guard = debugGuardBytes();
		WH_HIP( hipMalloc( &v, (size_t)( bytes + 2 * guard ) ) );
		uint8_t* const body = (uint8_t*)v + guard;
		if( guard )
		{
			WH_HIP( hipMemsetAsync( v, GUARD_BYTE, (size_t)guard, stream ) );
			WH_HIP( hipMemsetAsync( body + bytes, GUARD_BYTE, (size_t)guard, stream ) );
		}
		const int fill = ( init == DONT_CARE && debugPoison() >= 0 ) ? debugPoison() : 0;
		WH_HIP( hipMemsetAsync( body, fill, (size_t)bytes, stream ) );
		allocations.push_back( { v, body, bytes, name } );
		vram += bytes;
		p = (T*)body;
		return 0;
	}
	static constexpr int GUARD_BYTE = 0xA5;
	static int debugPoison()
	{
		static const int v = []() { const char* e = getenv( "WH_DEBUG_POISON" ); return ( e && *e ) ? (int)( strtol( e, nullptr, 0 ) & 0xFF ) : -1; }();
		return v;
	}
	static int64_t debugGuardBytes() { return debugPoison() >= 0 ? 65536 : 0; }
	// Guard regions intact? Violations go to stderr with the buffer's name and the first damaged offset (negative = before the body).
	int verifyGuards() const
	{
		const int64_t guard = debugGuardBytes();
		if( !guard ) return 0;
		int bad = 0;
		std::vector<uint8_t> h( (size_t)guard );
		for( const Allocation& a : allocations )
			for( int side = 0; side < 2; side++ )
			{
				const uint8_t* const src = side ? (const uint8_t*)a.body + a.bytes : (const uint8_t*)a.base;
				if( hipMemcpy( h.data(), src, (size_t)guard, hipMemcpyDeviceToHost ) != hipSuccess ) continue;
				for( int64_t i = 0; i < guard; i++ )
					if( h[ (size_t)i ] != GUARD_BYTE )
					{
						fprintf( stderr, "WH_GUARD_VIOLATION: buffer '%s' (%lld bytes): write at offset %lld\n", a.name, (long long)a.bytes,
							(long long)( side ? a.bytes + i : i - guard ) );
						bad++;
						break;
					}
			}
		return bad;
	}
};


// Debug capture: device-to-device copy of an intermediate into a lazily allocated side buffer (stream-ordered).
template<class T>
static int capture( wh_context* c, T*& dst, const T* src, int64_t count, int64_t capacity )
{
	if( !( c->flags & WH_FLAG_DEBUG_CAPTURE ) ) return 0;
	if( !dst ) WH_CHECK( c->alloc( dst, capacity, wh_context::DONT_CARE, "debug capture" ) );
	WH_HIP( hipMemcpyAsync( dst, src, (size_t)count * sizeof( T ), hipMemcpyDeviceToDevice, c->stream ) );
	return 0;
}

// WH_DEBUG_SYNC=1: every launch is announced on stderr and waited for (and nothing is captured into a hipGraph), so that a
// device fault can be attributed to a kernel class from the log of a dead process.
static bool debugSync()
{
	static const bool v = []() { const char* e = getenv( "WH_DEBUG_SYNC" ); return e && *e && *e != '0'; }();
	return v;
}

// Runs one launch, optionally bracketed by events. flops / bytes are the ALGORITHMIC work of the launch.
template<class F>
static int profiled( wh_context* c, int kc, double flops, double bytes, F&& launch )
{
	Profiler& p = c->prof;
	if( debugSync() )
	{
		// breadcrumbs: a `Memory access fault by GPU` kills the process, the last line on stderr then names the launch
		fprintf( stderr, "[wh] launch %s\n", kernelClassNames[ kc ] );
		fflush( stderr );
		const int rc = launch();
		const hipError_t e = hipStreamSynchronize( c->stream );
		if( e != hipSuccess ) return hipFail( e, kernelClassNames[ kc ], __FILE__, __LINE__ );
		return rc;
	}
	if( !p.on ) return launch();
	hipEvent_t a = p.get(), b = p.get();
	WH_HIP( hipEventRecord( a, c->stream ) );
	const int rc = launch();
	WH_HIP( hipEventRecord( b, c->stream ) );
	p.pending.push_back( { kc, a, b } );
	p.calls[ kc ]++;
	p.flops[ kc ] += flops;
	p.bytes[ kc ] += bytes;
	if( p.pending.size() >= 4096 ) p.resolve();
	return rc;
}
// contexts alive PER DEVICE: a model on another adapter of the same process is nobody's neighbour
static std::atomic<int> g_liveContextsDev[ 64 ];
static std::atomic<int>& liveContexts( const wh_model* m ) { return g_liveContextsDev[ m->device & 63 ]; }
// TUNE_ENC_SERIAL: the encoders of the contexts of one device form a chain -- an encoder starts when the previous one (of another
// context) has finished. Stream-ordered (hipStreamWaitEvent), the host never blocks. Two batches started together then run out of
// phase from the first round on: while one decodes (launch latencies, HBM), the other's encoder has the matrix cores.
namespace
{
	std::mutex g_encGateMx;
	struct EncGate { hipEvent_t last = nullptr; const wh_context* owner = nullptr; };
	EncGate g_encGate[ 64 ];
	constexpr int ENC_SERIAL_MIN_WINDOWS = 8;	  // a one-window context (a single stream, a loader) neither waits nor makes others wait
}
static int gemmP( wh_context* c, const GemmArgs& g, bool skinny, bool decoder = false )
{
	const bool sk = skinny && g.M <= 32;
	const double flops = 2.0 * g.M * g.N * g.K;
	// algorithmic bytes: each operand once + the output once (FP16 in, 2..4 bytes out)
	const double bytes = 2.0 * g.N * g.K + 2.0 * g.M * g.K + ( g.out32 ? 4.0 : 2.0 ) * g.M * g.N;
	// a stream with a CU mask (WH_ENC_CUS): the persistent tiled kernel sizes its grid to the CUs it may use
	GemmArgs gl = g;
	if( c->encCus > 0 ) gl.cuLimit = c->stream == c->encStream ? c->encCus : c->totalCus - c->encCus;
	// Several contexts alive (batches in flight on their own streams): a workgroup of the persistent product owns its CU for the
	// whole launch (160 KiB of LDS, every register), so with all CUs taken a 10 us decode launch of the neighbouring batch waits up
	// to 2 ms for one. Leaving 4 CUs per XCD free costs the product 12 % of its CUs and returns 3 % of the whole job
	// (7389 -> 7627 audio-s/s, profiles/r03_ab_variants.txt); a lone context keeps the whole chip.
	else if( liveContexts( c->m ).load( std::memory_order_relaxed ) > 1 && c->totalCus >= 128 )
	{
		static const int spare = []() { const char* e = getenv( "WH_GEMM_SPARE_CUS" ); const int v = e ? atoi( e ) : 32; return v >= 0 && v <= 128 ? v & ~7 : 32; }();
		gl.cuLimit = c->totalCus - spare;
	}
	return profiled( c, sk ? KC_GEMM_SKINNY : ( decoder ? KC_GEMM_DEC : KC_GEMM_TILED ), flops, bytes, [ & ]() { return sk ? launchGemmSkinny( gl, c->stream ) : launchGemm( gl, c->stream ); } );
}
static int lnP( wh_context* c, const float* x, const float* w, const float* b, f16* out, int rows, int d, bool decoder = false )
{
	return profiled( c, decoder ? KC_LAYER_NORM_DEC : KC_LAYER_NORM, 8.0 * rows * d, 6.0 * rows * d, [ & ]() { return launchLayerNorm( x, w, b, out, rows, d, c->stream ); } );
}
static int attnDecP( wh_context* c, const DecAttnArgs& a, int keysHint = -1 )
{
	// ALGORITHMIC work of one launch: every K and V row the queries can see, once per (K/V block, head) -- the rows of a
	// prompt step and the hypotheses of a window share them -- with the real key count (keysHint when the position lives
	// in device memory), plus q in and the attention rows out. With a fused query also the residual rows and the query weight once.
	const int group = a.group > 0 ? a.group : 1;
	const double keys = (double)( keysHint > 0 ? keysHint : a.nKeys );
	const double d = (double)a.H * HEAD_DIM;
	double bytes = 2.0 * 2.0 * ( a.batch / group ) * a.H * keys * HEAD_DIM + 2.0 * 2.0 * a.batch * a.nTok * d;
	double flops = 4.0 * a.batch * a.H * keys * HEAD_DIM * a.nTok;
	if( a.lnX )
	{
		bytes += 4.0 * a.batch * a.nTok * d + 2.0 * d * d;
		flops += 2.0 * a.batch * a.nTok * d * d;
	}
	return profiled( c, a.causal ? KC_ATTN_DEC : KC_ATTN_DEC_CROSS, flops, bytes, [ & ]() { return launchAttentionDec( a, c->stream ); } );
}

// ==================================================================================================================
// device
// ==================================================================================================================
extern "C" {

const char* wh_last_error( void ) { return g_lastError.c_str(); }

int wh_device_count( void )
{
	int n = 0;
	if( hipGetDeviceCount( &n ) != hipSuccess ) return 0;
	return n;
}

int wh_device_info( int device, char* name, size_t nameCap, uint64_t* totalMemBytes, int* computeUnits )
{
	hipDeviceProp_t p;
	WH_HIP( hipGetDeviceProperties( &p, device ) );
	if( name && nameCap ) snprintf( name, nameCap, "%s (%s)", p.name, p.gcnArchName );
	if( totalMemBytes ) *totalMemBytes = p.totalGlobalMem;
	if( computeUnits ) *computeUnits = p.multiProcessorCount;
	return 0;
}

int wh_device_set( int device )
{
	WH_HIP( hipSetDevice( device ) );
	return 0;
}

// ==================================================================================================================
// model
// ==================================================================================================================
int64_t wh_model_arena_bytes( const wh_hparams* hp )
{
	if( checkHparams( hp ) ) return -1;
	return makeLayout( *hp ).total;
}

int wh_model_create( const wh_hparams* hp, void* arenaDev, int alreadyFilled, wh_model** out )
{
	if( !out ) { setError( "out is null" ); return WH_E_INVALIDARG; }
	WH_CHECK( checkHparams( hp ) );
	int nDev = 0;
	if( hipGetDeviceCount( &nDev ) != hipSuccess || nDev <= 0 )
	{
		setError( "no HIP device: libwhisper_hip has no CPU fallback" );
		return WH_E_NO_DEVICE;
	}
	wh_model* m = new wh_model();
	m->hp = *hp;
	m->L = makeLayout( *hp );
	if( hipGetDevice( &m->device ) != hipSuccess ) m->device = 0;
	if( arenaDev )
	{
		m->arena = (uint8_t*)arenaDev;
		m->ownsArena = false;
	}
	else
	{
		void* p = nullptr;
		const hipError_t e = hipMalloc( &p, (size_t)m->L.total );
		if( e != hipSuccess ) { delete m; return hipFail( e, "hipMalloc(arena)", __FILE__, __LINE__ ); }
		m->arena = (uint8_t*)p;
		m->ownsArena = true;
	}
	if( alreadyFilled )
		m->finalized = true;
	else
	{
		const hipError_t e = hipMemset( m->arena, 0, (size_t)m->L.total );
		if( e != hipSuccess ) { wh_model_destroy( m ); return hipFail( e, "hipMemset(arena)", __FILE__, __LINE__ ); }
	}
	*out = m;
	return 0;
}

void wh_model_destroy( wh_model* m )
{
	if( !m ) return;
	(void)bindDevice( m );
	if( m->ownsArena && m->arena ) (void)hipFree( m->arena );
	delete m;
}

static int upload( wh_model* m, int64_t off, const void* src, int64_t bytes )
{
	WH_BIND( m );
	WH_HIP( hipMemcpy( m->arena + off, src, (size_t)bytes, hipMemcpyHostToDevice ) );
	return 0;
}

// Destination of one file tensor. kind: 0 = plain copy, 1 = conv weight (re-ordered), rows x cols is the expected shape.
struct Slot
{
	int64_t off = -1;
	int64_t rows = 0, cols = 0;	   // expected numpy shape (rows, cols); vectors have rows = 1
	bool f16 = false;
	int kind = 0;
	int convIc = 0;
};

static bool parseBlock( const std::string& name, const char* prefix, int& idx, std::string& rest )
{
	const size_t pl = strlen( prefix );
	if( name.compare( 0, pl, prefix ) != 0 ) return false;
	size_t p = pl;
	if( p >= name.size() || !isdigit( (unsigned char)name[ p ] ) ) return false;
	int v = 0;
	while( p < name.size() && isdigit( (unsigned char)name[ p ] ) ) v = v * 10 + ( name[ p++ ] - '0' );
	if( p >= name.size() || name[ p ] != '.' ) return false;
	idx = v;
	rest = name.substr( p + 1 );
	return true;
}

// Tensor name map: Whisper/Whisper/WhisperModel.cpp:63-162 == Whisper/source/whisper.cpp:774-940
static bool resolve( const wh_model* m, const std::string& name, Slot& s )
{
	const wh_hparams& hp = m->hp;
	const Layout& L = m->L;
	const int64_t d = hp.n_audio_state;
	auto mat = [ & ]( int64_t off, int64_t rows, int64_t cols ) { s.off = off; s.rows = rows; s.cols = cols; s.f16 = true; return true; };
	auto vec = [ & ]( int64_t off, int64_t n ) { s.off = off; s.rows = 1; s.cols = n; s.f16 = false; return true; };
	if( name == "encoder.positional_embedding" ) { s.off = L.encPe; s.rows = hp.n_audio_ctx; s.cols = d; s.f16 = false; return true; }
	if( name == "encoder.conv1.weight" ) { s.kind = 1; s.convIc = hp.n_mels; return mat( L.conv1w, d, 3ll * hp.n_mels ); }
	if( name == "encoder.conv1.bias" ) return vec( L.conv1b, d );
	if( name == "encoder.conv2.weight" ) { s.kind = 1; s.convIc = (int)d; return mat( L.conv2w, d, 3 * d ); }
	if( name == "encoder.conv2.bias" ) return vec( L.conv2b, d );
	if( name == "encoder.ln_post.weight" ) return vec( L.lnPostW, d );
	if( name == "encoder.ln_post.bias" ) return vec( L.lnPostB, d );
	if( name == "decoder.positional_embedding" ) { s.off = L.decPe; s.rows = hp.n_text_ctx; s.cols = d; s.f16 = false; return true; }
	if( name == "decoder.token_embedding.weight" ) return mat( L.te, hp.n_vocab, d );
	if( name == "decoder.ln.weight" ) return vec( L.decLnW, d );
	if( name == "decoder.ln.bias" ) return vec( L.decLnB, d );
	int il = 0;
	std::string r;
	if( parseBlock( name, "encoder.blocks.", il, r ) )
	{
		if( il >= hp.n_audio_layer ) return false;
		const EncLayer& e = L.enc[ il ];
		if( r == "attn_ln.weight" ) return vec( e.ln1w, d );
		if( r == "attn_ln.bias" ) return vec( e.ln1b, d );
		if( r == "attn.query.weight" ) return mat( e.wqkv, d, d );
		if( r == "attn.query.bias" ) return vec( e.bqkv, d );
		if( r == "attn.key.weight" ) return mat( e.wqkv + 2 * d * d, d, d );
		if( r == "attn.value.weight" ) return mat( e.wqkv + 4 * d * d, d, d );
		if( r == "attn.value.bias" ) return vec( e.bqkv + 8 * d, d );
		if( r == "attn.out.weight" ) return mat( e.wo, d, d );
		if( r == "attn.out.bias" ) return vec( e.bo, d );
		if( r == "mlp_ln.weight" ) return vec( e.ln2w, d );
		if( r == "mlp_ln.bias" ) return vec( e.ln2b, d );
		if( r == "mlp.0.weight" ) return mat( e.w1, 4 * d, d );
		if( r == "mlp.0.bias" ) return vec( e.b1, 4 * d );
		if( r == "mlp.2.weight" ) return mat( e.w2, d, 4 * d );
		if( r == "mlp.2.bias" ) return vec( e.b2, d );
		return false;
	}
	if( parseBlock( name, "decoder.blocks.", il, r ) )
	{
		if( il >= hp.n_text_layer ) return false;
		const DecLayer& e = L.dec[ il ];
		if( r == "attn_ln.weight" ) return vec( e.ln1w, d );
		if( r == "attn_ln.bias" ) return vec( e.ln1b, d );
		if( r == "attn.query.weight" ) return mat( e.wqkv, d, d );
		if( r == "attn.query.bias" ) return vec( e.bqkv, d );
		if( r == "attn.key.weight" ) return mat( e.wqkv + 2 * d * d, d, d );
		if( r == "attn.value.weight" ) return mat( e.wqkv + 4 * d * d, d, d );
		if( r == "attn.value.bias" ) return vec( e.bqkv + 8 * d, d );
		if( r == "attn.out.weight" ) return mat( e.wo, d, d );
		if( r == "attn.out.bias" ) return vec( e.bo, d );
		if( r == "cross_attn_ln.weight" ) return vec( e.lncw, d );
		if( r == "cross_attn_ln.bias" ) return vec( e.lncb, d );
		if( r == "cross_attn.query.weight" ) return mat( e.wcq, d, d );
		if( r == "cross_attn.query.bias" ) return vec( e.bcq, d );
		if( r == "cross_attn.key.weight" ) return mat( L.wcross + 2 * ( 2ll * il ) * d * d, d, d );
		if( r == "cross_attn.value.weight" ) return mat( L.wcross + 2 * ( 2ll * il + 1 ) * d * d, d, d );
		if( r == "cross_attn.value.bias" ) return vec( L.bcross + 4 * ( 2ll * il + 1 ) * d, d );
		if( r == "cross_attn.out.weight" ) return mat( e.wco, d, d );
		if( r == "cross_attn.out.bias" ) return vec( e.bco, d );
		if( r == "mlp_ln.weight" ) return vec( e.ln2w, d );
		if( r == "mlp_ln.bias" ) return vec( e.ln2b, d );
		if( r == "mlp.0.weight" ) return mat( e.w1, 4 * d, d );
		if( r == "mlp.0.bias" ) return vec( e.b1, 4 * d );
		if( r == "mlp.2.weight" ) return mat( e.w2, d, 4 * d );
		if( r == "mlp.2.bias" ) return vec( e.b2, d );
		return false;
	}
	return false;
}

static inline uint16_t f32ToF16Bits( float f )
{
	const _Float16 h = (_Float16)f;
	uint16_t u;
	memcpy( &u, &h, 2 );
	return u;
}
static inline float f16BitsToF32( uint16_t u )
{
	_Float16 h;
	memcpy( &h, &u, 2 );
	return (float)h;
}

int wh_model_set_tensor( wh_model* m, const char* name, int nDims, const int32_t* ne, int isF16, const void* data )
{
	if( !m || !name || !ne || !data || nDims < 1 || nDims > 3 ) { setError( "set_tensor: bad argument" ); return WH_E_INVALIDARG; }
	if( m->finalized ) { setError( "set_tensor: model already finalized" ); return WH_E_INVALIDARG; }
	Slot s;
	if( !resolve( m, name, s ) )
	{
		setError( std::string( "unknown tensor '" ) + name + "' in model file" );
		return WH_E_INVALIDARG;
	}
	if( m->loaded.count( name ) )
	{
		setError( std::string( "tensor '" ) + name + "' appears twice" );
		return WH_E_INVALIDARG;
	}
	int64_t count = 1;
	for( int i = 0; i < nDims; i++ ) count *= ne[ i ];
	if( count != s.rows * s.cols )
	{
		setError( std::string( "tensor '" ) + name + "' has wrong size in model file" );
		return WH_E_INVALIDARG;
	}
	// shape check: ne[0] is the contiguous dimension
	bool shapeOk;
	if( s.kind == 1 )
		shapeOk = nDims == 3 && ne[ 0 ] == 3 && ne[ 1 ] == s.convIc && ne[ 2 ] == s.rows;
	else if( s.rows == 1 )
		shapeOk = ne[ nDims - 1 ] == s.cols || ( nDims >= 1 && ne[ 0 ] == s.cols ) || ( nDims == 2 && ne[ 0 ] == 1 && ne[ 1 ] == s.cols );
	else
		shapeOk = nDims == 2 && ne[ 0 ] == s.cols && ne[ 1 ] == s.rows;
	if( !shapeOk )
	{
		setError( std::string( "tensor '" ) + name + "' has wrong shape in model file" );
		return WH_E_INVALIDARG;
	}

	if( s.kind == 1 )
	{
		// file: [out][in][3] (tap contiguous) -> ours: [out][tap * in + c], row padded with zeros (conv as an implicit GEMM)
		const int64_t ic = s.convIc, oc = s.rows;
		const int64_t kpad = ( s.off == m->L.conv1w ) ? conv1Kpad( m->hp ) : 3 * ic;
		std::vector<uint16_t> tmp( (size_t)( oc * kpad ), 0 );
		for( int64_t o = 0; o < oc; o++ )
			for( int64_t c = 0; c < ic; c++ )
				for( int t = 0; t < 3; t++ )
				{
					const int64_t si = ( o * ic + c ) * 3 + t;
					const uint16_t v = isF16 ? ( (const uint16_t*)data )[ si ] : f32ToF16Bits( ( (const float*)data )[ si ] );
					tmp[ (size_t)( o * kpad + t * ic + c ) ] = v;
				}
		WH_CHECK( upload( m, s.off, tmp.data(), (int64_t)tmp.size() * 2 ) );
	}
	else if( s.f16 )
	{
		if( isF16 )
			WH_CHECK( upload( m, s.off, data, count * 2 ) );
		else
		{
			std::vector<uint16_t> tmp( (size_t)count );
			for( int64_t i = 0; i < count; i++ ) tmp[ (size_t)i ] = f32ToF16Bits( ( (const float*)data )[ i ] );
			WH_CHECK( upload( m, s.off, tmp.data(), count * 2 ) );
		}
	}
	else
	{
		if( !isF16 )
			WH_CHECK( upload( m, s.off, data, count * 4 ) );
		else
		{
			std::vector<float> tmp( (size_t)count );
			for( int64_t i = 0; i < count; i++ ) tmp[ (size_t)i ] = f16BitsToF32( ( (const uint16_t*)data )[ i ] );
			WH_CHECK( upload( m, s.off, tmp.data(), count * 4 ) );
		}
	}
	m->loaded.insert( name );
	return 0;
}

int wh_model_set_filters( wh_model* m, int nMel, int nFft, const float* data )
{
	if( !m || !data ) { setError( "set_filters: bad argument" ); return WH_E_INVALIDARG; }
	if( nMel != m->hp.n_mels || nFft != 201 ) { setError( "mel filterbank must be [n_mels][201]" ); return WH_E_INVALIDARG; }
	WH_CHECK( upload( m, m->L.filters, data, 4ll * nMel * nFft ) );
	m->filtersSet = true;
	return 0;
}

int wh_model_finalize( wh_model* m )
{
	if( !m ) return WH_E_INVALIDARG;
	if( m->finalized ) return 0;
	if( !m->filtersSet )
	{
		// the reference's loader fails on a file without the filterbank (WhisperModel.cpp:446-456); an all-zero one would turn
		// every spectrogram into log10(1e-10) without an error
		setError( "mel filterbank has not been set (wh_model_set_filters)" );
		return WH_E_NOT_READY;
	}
	if( m->loaded.size() != m->expectedTensors() )
	{
		char buf[ 160 ];
		snprintf( buf, sizeof( buf ), "not all tensors loaded from model file - expected %zu, got %zu", m->expectedTensors(), m->loaded.size() );
		setError( buf );
		return WH_E_NOT_READY;
	}
	// DFT twiddles for the mel kernel: cos / sin of 2 pi n / 400 in double (same expression as whisper.cpp:2073-2077)
	std::vector<double> tw( 800 );
	for( int n = 0; n < 400; n++ )
	{
		tw[ n ] = cos( ( 2.0 * M_PI * n ) / 400 );
		tw[ 400 + n ] = sin( ( 2.0 * M_PI * n ) / 400 );
	}
	WH_CHECK( upload( m, m->L.dft, tw.data(), 800 * 8 ) );
	// The reference's exponential IS a table: table_exp_f16[ bits ] = fp16( expf( fp32( fp16 bits ) ) ), built once at start-up
	// (Whisper/source/ggml.c:1375-1385, read at :5069-5080 and :6001-6016). The softmax only ever looks up non-positive arguments, and
	// fp16( expf( x ) ) is 0 below -17.33: entry i here belongs to the FP16 number -|bits i|, i < 0x5000 (entries from 0x4C56 on are 0).
	// Built with the host's expf like the reference builds its own -- all 20480 entries equal the reference's table
	// (tests/test_gpu_ops.py::test_exp_table_in_the_arena). attentionEncT keeps it in LDS: 40 KB next to the K / V tiles.
	{
		std::vector<_Float16> tab( EXP_TABLE_ENTRIES );
		for( uint32_t i = 0; i < (uint32_t)EXP_TABLE_ENTRIES; i++ )
		{
			const uint16_t bits = (uint16_t)( i | 0x8000u );
			_Float16 h;
			memcpy( &h, &bits, 2 );
			tab[ i ] = (_Float16)expf( (float)h );
		}
		WH_CHECK( upload( m, m->L.expTab, tab.data(), EXP_TABLE_ENTRIES * 2 ) );
	}
	m->finalized = true;
	return 0;
}

int wh_model_arena( wh_model* m, void** dev, int64_t* bytes )
{
	if( !m ) return WH_E_INVALIDARG;
	if( dev ) *dev = m->arena;
	if( bytes ) *bytes = m->L.total;
	return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// RCCL: the arena of rank `root` into every rank's arena (one process per GPU). librccl.so is opened on first use.
// ------------------------------------------------------------------------------------------------------------------
namespace
{
	struct RcclApi
	{
		void* lib = nullptr;
		int ( *getUniqueId )( void* ) = nullptr;
		int ( *commInitRank )( void**, int, ncclUniqueIdBlob, int ) = nullptr;
		int ( *commDestroy )( void* ) = nullptr;
		int ( *broadcast )( const void*, void*, size_t, int, int, void*, hipStream_t ) = nullptr;
		int ( *allReduce )( const void*, void*, size_t, int, int, void*, hipStream_t ) = nullptr;
		const char* ( *errorString )( int ) = nullptr;
		int ( *getVersion )( int* ) = nullptr;	   // optional
		std::string why, loadedAs;
	};
	RcclApi* rccl()
	{
		static RcclApi api;
		static std::once_flag once;
		std::call_once( once, []() {
			const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
			for( const char* n : names )
			{
				api.lib = dlopen( n, RTLD_NOW | RTLD_LOCAL );
				if( api.lib ) { api.loadedAs = n; break; }
			}
			if( !api.lib )
			{
				const char* e = dlerror();
				api.why = std::string( "librccl.so not found: " ) + ( e ? e : "" );
				return;
			}
			auto sym = [ & ]( const char* name ) -> void* {
				void* p = dlsym( api.lib, name );
				if( !p && api.why.empty() ) api.why = std::string( "librccl.so lacks " ) + name;
				return p;
			};
			api.getUniqueId = (decltype( api.getUniqueId ))sym( "ncclGetUniqueId" );
			api.commInitRank = (decltype( api.commInitRank ))sym( "ncclCommInitRank" );
			api.commDestroy = (decltype( api.commDestroy ))sym( "ncclCommDestroy" );
			api.broadcast = (decltype( api.broadcast ))sym( "ncclBroadcast" );
			api.allReduce = (decltype( api.allReduce ))sym( "ncclAllReduce" );
			api.errorString = (decltype( api.errorString ))sym( "ncclGetErrorString" );
			api.getVersion = (decltype( api.getVersion ))dlsym( api.lib, "ncclGetVersion" );
		} );
		return &api;
	}
	int rcclFail( int rc, const char* what )
	{
		RcclApi* r = rccl();
		setError( std::string( what ) + ": " + ( r->errorString ? r->errorString( rc ) : "RCCL error" ) );
		return WH_E_HIP;
	}
}	// namespace

struct wh_comm
{
	void* comm = nullptr;
	int rank = 0, world = 1;
	hipStream_t stream = nullptr;
	int* scratch = nullptr;
	double timeout = 0.0;	   // seconds a collective may take before the call gives up (0 = wait for ever)
};

namespace
{
	// Waits for the communicator's stream like hipStreamSynchronize, but not for ever: a rank that never arrives at a collective
	// must turn into an error on the ranks that did, not into a hung node (RCCL itself has no deadline).
	int waitComm( wh_comm* c, const char* what )
	{
		if( c->timeout <= 0.0 )
		{
			WH_HIP( hipStreamSynchronize( c->stream ) );
			return 0;
		}
		const auto t0 = std::chrono::steady_clock::now();
		for( long spins = 0;; spins++ )
		{
			const hipError_t e = hipStreamQuery( c->stream );
			if( e == hipSuccess ) return 0;
			if( e != hipErrorNotReady ) return hipFail( e, what, __FILE__, __LINE__ );
			(void)hipGetLastError();
			if( std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count() > c->timeout )
			{
				char buf[ 160 ];
				snprintf( buf, sizeof( buf ), "%s: rank %d of %d gave up after %.0f s (a rank did not arrive)", what, c->rank, c->world, c->timeout );
				setError( buf );
				return WH_E_TIMEOUT;
			}
			if( spins > 64 ) std::this_thread::sleep_for( std::chrono::microseconds( 200 ) );
		}
	}
}

// Packaging check, callable without a GPU: is the collective library there under one of the names wh_comm_* opens, with every entry point they use?
// (The first multi-GPU lease must not fail before its first timed step for a reason a single-GPU box could have shown.)
int wh_comm_runtime_check( char* detail, size_t detailCap )
{
	RcclApi* r = rccl();
	const bool ok = r->lib && r->why.empty() && r->getUniqueId && r->commInitRank && r->commDestroy && r->broadcast && r->allReduce && r->errorString;
	int version = 0;
	if( ok && r->getVersion ) (void)r->getVersion( &version );
	if( detail && detailCap )
	{
		if( ok ) snprintf( detail, detailCap, "opened as %s, version %d, ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclBroadcast / ncclAllReduce / ncclGetErrorString resolved",
			r->loadedAs.c_str(), version );
		else snprintf( detail, detailCap, "%s", r->why.empty() ? "librccl.so: an entry point is missing" : r->why.c_str() );
	}
	if( !ok ) { setError( r->why.empty() ? "librccl.so: an entry point is missing" : r->why ); return WH_E_NOT_READY; }
	return 0;
}

int wh_comm_unique_id( void* id128 )
{
	if( !id128 ) { setError( "comm_unique_id: null argument" ); return WH_E_INVALIDARG; }
	RcclApi* r = rccl();
	if( !r->getUniqueId ) { setError( "comm_unique_id: " + r->why ); return WH_E_NO_DEVICE; }
	const int rc = r->getUniqueId( id128 );
	return rc == 0 ? 0 : rcclFail( rc, "ncclGetUniqueId" );
}

int wh_comm_create( const void* id128, int rank, int worldSize, wh_comm** out )
{
	return wh_comm_create_timeout( id128, rank, worldSize, 0.0, out );
}

int wh_comm_create_timeout( const void* id128, int rank, int worldSize, double timeoutSeconds, wh_comm** out )
{
	if( !id128 || !out || worldSize <= 0 || rank < 0 || rank >= worldSize ) { setError( "comm_create: bad argument" ); return WH_E_INVALIDARG; }
	RcclApi* r = rccl();
	if( !r->commInitRank || !r->broadcast || !r->commDestroy ) { setError( "comm_create: " + r->why ); return WH_E_NO_DEVICE; }
	wh_comm* c = new wh_comm();
	c->rank = rank; c->world = worldSize;
	c->timeout = timeoutSeconds > 0.0 ? timeoutSeconds : 0.0;
	ncclUniqueIdBlob id;
	memcpy( id.internal, id128, WH_COMM_ID_BYTES );
	int rc = 0;
	if( c->timeout <= 0.0 )
		rc = r->commInitRank( &c->comm, worldSize, id, rank );		// uses the calling thread's current device
	else
	{
		// ncclCommInitRank blocks until every rank has called it. With a deadline it runs on a helper thread (bound to the caller's
		// device); when the deadline passes the call returns WH_E_TIMEOUT and the helper is left behind -- the process is expected to
		// exit (whisper-mgpu does), nothing else can be done with a rendezvous that never completes.
		int device = 0;
		WH_HIP( hipGetDevice( &device ) );
		struct Rendezvous { std::mutex mx; std::condition_variable cv; bool done = false; int rc = 0; void* comm = nullptr; };
		auto rv = std::make_shared<Rendezvous>();
		std::thread( [ rv, r, worldSize, id, rank, device ]() {
			(void)hipSetDevice( device );
			void* comm = nullptr;
			const int rcInit = r->commInitRank( &comm, worldSize, id, rank );
			std::lock_guard<std::mutex> lk( rv->mx );
			rv->rc = rcInit; rv->comm = comm; rv->done = true;
			rv->cv.notify_all();
		} ).detach();
		std::unique_lock<std::mutex> lk( rv->mx );
		if( !rv->cv.wait_for( lk, std::chrono::duration<double>( c->timeout ), [ & ]() { return rv->done; } ) )
		{
			char buf[ 160 ];
			snprintf( buf, sizeof( buf ), "ncclCommInitRank: rank %d of %d gave up after %.0f s (a rank did not arrive)", rank, worldSize, c->timeout );
			setError( buf );
			delete c;
			return WH_E_TIMEOUT;
		}
		rc = rv->rc;
		c->comm = rv->comm;
	}
	if( rc != 0 ) { delete c; return rcclFail( rc, "ncclCommInitRank" ); }
	hipError_t e = hipStreamCreateWithFlags( &c->stream, hipStreamNonBlocking );
	if( e == hipSuccess ) e = hipMalloc( (void**)&c->scratch, 8 );
	if( e == hipSuccess ) e = hipMemset( c->scratch, 0, 8 );
	if( e != hipSuccess ) { (void)wh_comm_destroy( c ); return hipFail( e, "comm_create", __FILE__, __LINE__ ); }
	*out = c;
	return 0;
}

int wh_comm_destroy( wh_comm* c )
{
	if( !c ) return 0;
	if( c->stream ) (void)hipStreamSynchronize( c->stream );
	if( c->comm && rccl()->commDestroy ) (void)rccl()->commDestroy( c->comm );
	if( c->scratch ) (void)hipFree( c->scratch );
	if( c->stream ) (void)hipStreamDestroy( c->stream );
	delete c;
	return 0;
}

int wh_comm_info( const wh_comm* c, int* rank, int* worldSize )
{
	if( !c ) { setError( "comm_info: null communicator" ); return WH_E_INVALIDARG; }
	if( rank ) *rank = c->rank;
	if( worldSize ) *worldSize = c->world;
	return 0;
}

int wh_comm_barrier( wh_comm* c )
{
	if( !c ) { setError( "comm_barrier: null communicator" ); return WH_E_INVALIDARG; }
	RcclApi* r = rccl();
	if( !r->allReduce ) { setError( "comm_barrier: " + r->why ); return WH_E_NO_DEVICE; }
	const int rc = r->allReduce( c->scratch, c->scratch + 1, 1, 2 /* ncclInt32 */, 0 /* ncclSum */, c->comm, c->stream );
	if( rc != 0 ) return rcclFail( rc, "ncclAllReduce" );
	return waitComm( c, "comm_barrier" );
}

int wh_comm_set_timeout( wh_comm* c, double seconds )
{
	if( !c ) { setError( "comm_set_timeout: null communicator" ); return WH_E_INVALIDARG; }
	c->timeout = seconds > 0.0 ? seconds : 0.0;
	return 0;
}

int wh_comm_broadcast_i32( wh_comm* c, int root, int32_t* value )
{
	if( !c || !value || root < 0 || root >= c->world ) { setError( "comm_broadcast_i32: bad argument" ); return WH_E_INVALIDARG; }
	RcclApi* r = rccl();
	if( !r->broadcast ) { setError( "comm_broadcast_i32: " + r->why ); return WH_E_NO_DEVICE; }
	if( c->rank == root ) WH_HIP( hipMemcpyAsync( c->scratch, value, 4, hipMemcpyHostToDevice, c->stream ) );
	const int rc = r->broadcast( c->scratch, c->scratch, 4, 0 /* ncclInt8 */, root, c->comm, c->stream );
	if( rc != 0 ) return rcclFail( rc, "ncclBroadcast" );
	WH_CHECK( waitComm( c, "comm_broadcast_i32" ) );
	WH_HIP( hipMemcpy( value, c->scratch, 4, hipMemcpyDeviceToHost ) );
	WH_HIP( hipMemset( c->scratch, 0, 8 ) );
	return 0;
}

int wh_model_broadcast( wh_model* m, wh_comm* c, int root, double* secondsOut )
{
	if( !m || !c || root < 0 || root >= c->world ) { setError( "model_broadcast: bad argument" ); return WH_E_INVALIDARG; }
	if( c->rank == root && !m->finalized ) { setError( "model_broadcast: the root's model is not finalized" ); return WH_E_NOT_READY; }
	WH_BIND( m );
	RcclApi* r = rccl();
	const int64_t bytes = wh_model_arena_bytes( &m->hp );
	WH_HIP( hipDeviceSynchronize() );	   // the root's uploads (synchronous copies on the null stream) are behind us on every stream
	const auto t0 = std::chrono::steady_clock::now();
	const int rc = r->broadcast( m->arena, m->arena, (size_t)bytes, 0 /* ncclInt8 */, root, c->comm, c->stream );
	if( rc != 0 ) return rcclFail( rc, "ncclBroadcast" );
	WH_CHECK( waitComm( c, "model_broadcast" ) );
	if( secondsOut ) *secondsOut = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
	if( c->rank != root )
	{
		// the image holds everything wh_model_finalize builds on the root (derived tables live in the arena)
		m->finalized = true;
		m->filtersSet = true;
	}
	return 0;
}

int wh_model_hparams( const wh_model* m, wh_hparams* out )
{
	if( !m || !out ) return WH_E_INVALIDARG;
	*out = m->hp;
	return 0;
}

// ==================================================================================================================
// context
// ==================================================================================================================
int wh_context_create( wh_model* m, int maxBatch, void* stream, wh_context** out )
{
	return wh_context_create_hyp( m, maxBatch, 1, stream, out );
}

int wh_context_create_hyp( wh_model* m, int maxBatch, int hypotheses, void* stream, wh_context** out )
{
	if( !m || !out || maxBatch <= 0 || hypotheses <= 0 || hypotheses > 8 || hypotheses == 6 || hypotheses == 7 )
	{
		setError( "context_create: bad argument (hypotheses per window: 1, 2, 3, 4, 5 or 8)" );
		return WH_E_INVALIDARG;
	}
	if( !m->finalized ) { setError( "context_create: model is not finalized" ); return WH_E_NOT_READY; }
	WH_BIND( m );
	wh_context* c = new wh_context();
	liveContexts( m ).fetch_add( 1 );
	c->m = m;
	(void)hipDeviceGetAttribute( &c->totalCus, hipDeviceAttributeMultiprocessorCount, m->device );
	c->maxBatch = maxBatch;
	c->hyp = hypotheses;
	c->maxSeq = maxBatch * hypotheses;
	c->stream = (hipStream_t)stream;
	if( !c->stream )
	{
		// the legacy null stream cannot be captured into a hipGraph: own a non-blocking stream instead
		hipError_t e;
		if( g_tuning & TUNE_SPLIT_STREAMS )
		{
			int lo = 0, hi = 0;
			(void)hipDeviceGetStreamPriorityRange( &lo, &hi );	   // lo = least, hi = greatest priority (numerically lower)
			// WH_ENC_CUS = n: SPATIAL split instead of priorities -- the encoder stream may use n CUs, the decode stream the others.
			// The persistent encoder GEMM takes a whole CU (160 KiB of LDS, every register), so two batches in flight otherwise
			// take turns; with disjoint CU sets the MFMA-bound encoder of one batch runs beside the HBM-bound decode chain of the
			// other. Mask bit i is a CU of XCD i % 8 (the driver spreads a queue's mask over the XCDs), so the low n bits give the
			// encoder n / 8 CUs of EVERY XCD and both sides keep all eight L2s and fabric links.
			int encCus = 0, cus = 0;
			if( const char* ev = getenv( "WH_ENC_CUS" ) ) encCus = atoi( ev );
			(void)hipDeviceGetAttribute( &cus, hipDeviceAttributeMultiprocessorCount, m->device );
			if( encCus >= 8 && encCus <= cus - 8 )
			{
				encCus &= ~7;
				uint32_t maskE[ 16 ] = {}, maskD[ 16 ] = {};
				for( int i = 0; i < cus && i < 512; i++ ) ( i < encCus ? maskE : maskD )[ i >> 5 ] |= 1u << ( i & 31 );
				const uint32_t words = (uint32_t)( ( cus + 31 ) / 32 );
				e = hipExtStreamCreateWithCUMask( &c->stream, words, maskD );
				if( e == hipSuccess ) e = hipExtStreamCreateWithCUMask( &c->encStream, words, maskE );
				c->encCus = encCus;
				c->totalCus = cus;
			}
			else
			{
				e = hipStreamCreateWithPriority( &c->stream, hipStreamNonBlocking, hi );
				if( e == hipSuccess ) e = hipStreamCreateWithPriority( &c->encStream, hipStreamNonBlocking, lo );
			}
			if( e == hipSuccess ) e = hipEventCreateWithFlags( &c->encReady, hipEventDisableTiming );
			if( e == hipSuccess ) e = hipEventCreateWithFlags( &c->encDone, hipEventDisableTiming );
		}
		else
			e = hipStreamCreateWithFlags( &c->stream, hipStreamNonBlocking );
		c->ownsStream = true;
		if( e != hipSuccess ) { wh_context_destroy( c ); return hipFail( e, "hipStreamCreate", __FILE__, __LINE__ ); }	// releases whichever streams / events exist
	}
	const wh_hparams& hp = m->hp;
	const int64_t d = hp.n_audio_state, B = maxBatch, H = hp.n_audio_head, S = c->maxSeq;
	const int T = hp.n_audio_ctx;
	c->T = T;
	c->Tpad = roundUp( T, 256 );
	c->maxRows = c->maxSeq * hp.n_text_ctx;
	{
		// option enc_chunk = the most windows one encoder pass takes (default 128); a larger batch is cut into equal chunks
		const int chunkMax = g_opt.encChunk >= 1 && g_opt.encChunk <= 1024 ? g_opt.encChunk : 128;
		const int nChunks = ( maxBatch + chunkMax - 1 ) / chunkMax;
		c->encChunk = ( maxBatch + nChunks - 1 ) / nChunks;
	}
	const int64_t Be = c->encChunk;
	const int64_t rowsE = Be * T;
	c->convInStride = ( 2ll * T + 2 ) * hp.n_mels;
	c->conv1Stride = ( 2ll * T + 2 ) * d;
	int rc = 0;
	// conv1 is an implicit GEMM whose row t is the K = conv1Kpad halfs starting at padded row t (stride n_mels): the last
	// row of the last window therefore reads conv1Kpad - 3 n_mels halfs (16 at 80 mels, 0 at 128) past the logical end. Those
	// columns meet zero weights (the padded part of conv1w's rows), but the operand must still be finite: the tail is part
	// of the allocation and, like the padding rows, stays zero for the life of the context. conv2 (K = 3 d over rows of
	// stride 2 d, last row ending at (2 T + 1) d) never leaves its (2 T + 2) d rows.
	const int64_t conv1Tail = conv1Kpad( hp ) - 3 * hp.n_mels;
	rc = rc ? rc : c->alloc( c->convIn, Be * c->convInStride + conv1Tail, wh_context::MUST_BE_ZERO, "convIn" );
	rc = rc ? rc : c->alloc( c->conv1Out, Be * c->conv1Stride, wh_context::MUST_BE_ZERO, "conv1Out" );
	rc = rc ? rc : c->alloc( c->x, rowsE * d, wh_context::DONT_CARE, "x" );
	rc = rc ? rc : c->alloc( c->xn, rowsE * d, wh_context::DONT_CARE, "xn" );
	rc = rc ? rc : c->alloc( c->q, rowsE * d, wh_context::DONT_CARE, "q" );
	rc = rc ? rc : c->alloc( c->k, rowsE * d, wh_context::DONT_CARE, "k" );
	rc = rc ? rc : c->alloc( c->vT, Be * H * HEAD_DIM * c->Tpad, wh_context::MUST_BE_ZERO, "vT" );
	rc = rc ? rc : c->alloc( c->attn, rowsE * d, wh_context::DONT_CARE, "attn" );
	rc = rc ? rc : c->alloc( c->h, rowsE * 4 * d, wh_context::DONT_CARE, "h" );
	rc = rc ? rc : c->alloc( c->crossK, (int64_t)hp.n_text_layer * B * T * d, wh_context::MUST_BE_ZERO, "crossK" );
	rc = rc ? rc : c->alloc( c->crossV, (int64_t)hp.n_text_layer * B * T * d, wh_context::MUST_BE_ZERO, "crossV" );
	rc = rc ? rc : c->alloc( c->selfK, (int64_t)hp.n_text_layer * S * hp.n_text_ctx * d, wh_context::MUST_BE_ZERO, "selfK" );
	rc = rc ? rc : c->alloc( c->selfV, (int64_t)hp.n_text_layer * S * hp.n_text_ctx * d, wh_context::MUST_BE_ZERO, "selfV" );
	const int64_t rowsD = c->maxRows;
	rc = rc ? rc : c->alloc( c->dx, rowsD * d, wh_context::DONT_CARE, "dx" );
	rc = rc ? rc : c->alloc( c->dxn, rowsD * d, wh_context::DONT_CARE, "dxn" );
	rc = rc ? rc : c->alloc( c->dq, rowsD * d, wh_context::DONT_CARE, "dq" );
	rc = rc ? rc : c->alloc( c->dattn, rowsD * d, wh_context::DONT_CARE, "dattn" );
	rc = rc ? rc : c->alloc( c->dh, rowsD * 4 * d, wh_context::DONT_CARE, "dh" );
	rc = rc ? rc : c->alloc( c->splitK, 8ll * ( rowsD < GEMV_FUSED_MAX_ROWS ? rowsD : GEMV_FUSED_MAX_ROWS ) * d, wh_context::DONT_CARE, "splitK" );	   // partial tiles of the K-split MLP down-projection (33 .. 128 rows)
	{
		const int64_t sm = S < SMALL_MAX_ROWS ? S : SMALL_MAX_ROWS;
		rc = rc ? rc : c->alloc( c->crossScores, sm * hp.n_text_head * T, wh_context::DONT_CARE, "crossScores" );
		rc = rc ? rc : c->alloc( c->crossSplitMax, sm * hp.n_text_head * CROSS_SPLITS, wh_context::DONT_CARE, "crossSplitMax" );
		rc = rc ? rc : c->alloc( c->crossPart, sm * hp.n_text_head * CROSS_SPLITS * CROSS_PART, wh_context::DONT_CARE, "crossPart" );
	}
	rc = rc ? rc : c->alloc( c->logits, S * (int64_t)hp.n_vocab, wh_context::DONT_CARE, "logits" );
	rc = rc ? rc : c->alloc( c->probs, S * (int64_t)hp.n_vocab, wh_context::DONT_CARE, "probs" );
	rc = rc ? rc : c->alloc( c->tokensDev, rowsD, wh_context::DONT_CARE, "tokensDev" );
	rc = rc ? rc : c->alloc( c->melOffsetsDev, B, wh_context::DONT_CARE, "melOffsetsDev" );
	rc = rc ? rc : c->alloc( c->melWindowsDev, B, wh_context::MUST_BE_ZERO, "melWindowsDev" );
	rc = rc ? rc : c->alloc( c->tokDataDev, S, wh_context::DONT_CARE, "tokDataDev" );
	rc = rc ? rc : c->alloc( c->melScratch, 64 + 4 * MEL_BATCH_MAX, wh_context::DONT_CARE, "melScratch" );	  // [0..15]: the single / streamed entry points, then one maximum per buffer of a batch
	rc = rc ? rc : c->alloc( c->state, 1, wh_context::MUST_BE_ZERO, "state" );
	rc = rc ? rc : c->alloc( c->seqPos, S, wh_context::MUST_BE_ZERO, "seqPos" );
	if( rc == 0 )
	{
		c->pinnedInts = wh_context::PIN_POS + S + S * (int64_t)hp.n_text_ctx;
		const hipError_t e = hipHostMalloc( (void**)&c->pinned, sizeof( int32_t ) * (size_t)c->pinnedInts, hipHostMallocDefault );
		if( e != hipSuccess ) rc = hipFail( e, "hipHostMalloc", __FILE__, __LINE__ );
	}
	rc = rc ? rc : c->alloc( c->greedyOut, (int64_t)hp.n_text_ctx * S, wh_context::DONT_CARE, "greedyOut" );
	if( rc == 0 && !getenv( "WH_NO_MAILBOX" ) )
	{
		// optional: without it (allocation refused, WH_NO_MAILBOX) wh_decode_window_fetch waits for an event and copies
		const size_t n = (size_t)hp.n_text_ctx * S;
		void *d = nullptr, *f = nullptr;
		if( hipHostMalloc( &d, n * sizeof( TokenData ), hipHostMallocMapped | hipHostMallocCoherent ) == hipSuccess &&
			hipHostMalloc( &f, 2 * n * sizeof( int ), hipHostMallocMapped | hipHostMallocCoherent ) == hipSuccess )
		{
			memset( f, 0, 2 * n * sizeof( int ) );	   // stamp + checksum per record
			void *dd = nullptr, *fd = nullptr;
			if( hipHostGetDevicePointer( &dd, d, 0 ) == hipSuccess && hipHostGetDevicePointer( &fd, f, 0 ) == hipSuccess )
			{
				c->mailData = (TokenData*)d; c->mailFlag = (int*)f;
				c->mailDev = { (TokenData*)dd, (int*)fd };
				d = f = nullptr;
			}
		}
		if( d ) (void)hipHostFree( d );
		if( f ) (void)hipHostFree( f );
		(void)hipGetLastError();
	}
	if( rc == 0 )
	{
		const hipError_t e = hipStreamSynchronize( c->stream );
		if( e != hipSuccess ) rc = hipFail( e, "hipStreamSynchronize", __FILE__, __LINE__ );
	}
	if( rc != 0 )
	{
		wh_context_destroy( c );
		return rc;
	}
	*out = c;
	return 0;
}

void wh_context_destroy( wh_context* c )
{
	if( !c ) return;
	liveContexts( c->m ).fetch_sub( 1 );
	(void)bindDevice( c->m );
	if( c->stream ) (void)hipStreamSynchronize( c->stream );
	if( c->graphExec ) (void)hipGraphExecDestroy( c->graphExec );
	if( c->beamGraphExec ) (void)hipGraphExecDestroy( c->beamGraphExec );
	for( auto& mk : c->marks ) (void)hipEventDestroy( mk.ev );
	for( hipEvent_t e : c->markPool ) (void)hipEventDestroy( e );
	if( c->copyStream ) (void)hipStreamDestroy( c->copyStream );
	if( c->encStream ) { (void)hipStreamSynchronize( c->encStream ); (void)hipStreamDestroy( c->encStream ); }
	if( c->encReady ) (void)hipEventDestroy( c->encReady );
	if( c->encDone ) (void)hipEventDestroy( c->encDone );
	{
		std::lock_guard<std::mutex> lk( g_encGateMx );
		for( EncGate& gate : g_encGate )
			if( gate.owner == c ) gate = EncGate{};
	}
	if( c->encGateEv ) (void)hipEventDestroy( c->encGateEv );
	if( c->verifyGuards() != 0 ) fprintf( stderr, "WH_GUARD_VIOLATION: context %p wrote outside its buffers\n", (void*)c );
	for( const auto& a : c->allocations ) (void)hipFree( a.base );
	for( void* p : c->ex.owned ) (void)hipFree( p );
	if( c->pinned ) (void)hipHostFree( c->pinned );
	if( c->mailData ) (void)hipHostFree( c->mailData );
	if( c->mailFlag ) (void)hipHostFree( c->mailFlag );
	if( c->ownsStream && c->stream ) (void)hipStreamDestroy( c->stream );
	delete c;
}

int wh_context_bind( wh_context* c )
{
	if( !c ) return WH_E_INVALIDARG;
	WH_BIND( c->m );
	return 0;
}

int wh_context_set_audio_ctx( wh_context* c, int audioCtx )
{
	if( !c ) return WH_E_INVALIDARG;
	const wh_hparams& hp = c->m->hp;
	const int T = audioCtx > 0 ? audioCtx : hp.n_audio_ctx;
	if( T > hp.n_audio_ctx ) { setError( "audio_ctx exceeds the model's n_audio_ctx" ); return WH_E_INVALIDARG; }
	if( T == c->T ) return 0;
	WH_BIND( c->m );
	WH_HIP( hipStreamSynchronize( c->stream ) );
	// every launch reads the key count from the context (row strides of the caches included), so the override is the context's T; what was captured or
	// encoded with another T is void
	if( c->graphExec ) { (void)hipGraphExecDestroy( c->graphExec ); c->graphExec = nullptr; c->graphBatch = 0; }
	if( c->beamGraphExec ) { (void)hipGraphExecDestroy( c->beamGraphExec ); c->beamGraphExec = nullptr; c->beamGraphBatch = 0; }
	c->T = T;
	c->Tpad = roundUp( T, 256 );
	c->encoded = false;
	// the convolutions' zero padding sits right behind the last frame: rows 2 T + 1 of the conv input and of conv1's output may hold a longer window's data
	for( int b = 0; b < c->encChunk; b++ )
	{
		WH_HIP( hipMemsetAsync( c->convIn + b * c->convInStride + ( 2ll * T + 1 ) * hp.n_mels, 0, (size_t)hp.n_mels * 2, c->stream ) );
		WH_HIP( hipMemsetAsync( c->conv1Out + b * c->conv1Stride + ( 2ll * T + 1 ) * hp.n_audio_state, 0, (size_t)hp.n_audio_state * 2, c->stream ) );
	}
	// ... and the V operand's key padding [T, Tpad) must be finite and is expected to be zero
	WH_HIP( hipMemsetAsync( c->vT, 0, (size_t)c->encChunk * hp.n_audio_head * HEAD_DIM * roundUp( hp.n_audio_ctx, 256 ) * 2, c->stream ) );
	return 0;
}

int wh_context_set_flags( wh_context* c, uint32_t flags, int parityThreads )
{
	if( !c ) return WH_E_INVALIDARG;
	c->flags = flags;
	// WH_FLAG_PARITY_EXACT with 0 threads: the decoder's P.V as ONE correctly rounded sum per output (double accumulation) instead of the reference's
	// FP16 accumulation -- the thread-count-independent value every thread count of the reference approximates
	c->parityThreads = parityThreads > 0 ? parityThreads : ( ( flags & WH_FLAG_PARITY_EXACT ) && parityThreads == 0 ? 0 : 1 );
	return 0;
}

// ---- plain device buffers for host code that must not include HIP headers (replaces Whisper/D3D/createBuffer.cpp) ----
// With WH_DEBUG_POISON set these buffers get the same treatment as the context's own: poison fill, guard regions on both
// sides, guards verified by wh_buffer_free.
namespace
{
	struct GuardedBuffer { void* base; int64_t bytes; };
	std::mutex g_guardedMutex;
	std::map<void*, GuardedBuffer> g_guarded;
}
int wh_buffer_alloc( int64_t bytes, void** dev )
{
	if( !dev || bytes <= 0 ) { setError( "buffer_alloc: bad argument" ); return WH_E_INVALIDARG; }
	void* p = nullptr;
	const int64_t guard = wh_context::debugGuardBytes();
	const hipError_t e = hipMalloc( &p, (size_t)( bytes + 2 * guard ) );
	if( e != hipSuccess ) { hipFail( e, "hipMalloc", __FILE__, __LINE__ ); return e == hipErrorOutOfMemory ? WH_E_OUTOFMEMORY : WH_E_HIP; }
	if( guard )
	{
		uint8_t* const body = (uint8_t*)p + guard;
		WH_HIP( hipMemset( p, wh_context::GUARD_BYTE, (size_t)guard ) );
		WH_HIP( hipMemset( body, wh_context::debugPoison(), (size_t)bytes ) );
		WH_HIP( hipMemset( body + bytes, wh_context::GUARD_BYTE, (size_t)guard ) );
		std::lock_guard<std::mutex> lock( g_guardedMutex );
		g_guarded[ body ] = { p, bytes };
		p = body;
	}
	*dev = p;
	return 0;
}

int wh_buffer_free( void* dev )
{
	if( !dev ) return 0;
	if( wh_context::debugGuardBytes() )
	{
		GuardedBuffer g = { nullptr, 0 };
		{
			std::lock_guard<std::mutex> lock( g_guardedMutex );
			auto it = g_guarded.find( dev );
			if( it != g_guarded.end() ) { g = it->second; g_guarded.erase( it ); }
		}
		if( g.base )
		{
			const int64_t guard = wh_context::debugGuardBytes();
			std::vector<uint8_t> h( (size_t)guard );
			for( int side = 0; side < 2; side++ )
			{
				const uint8_t* const src = side ? (const uint8_t*)dev + g.bytes : (const uint8_t*)g.base;
				if( hipMemcpy( h.data(), src, (size_t)guard, hipMemcpyDeviceToHost ) != hipSuccess ) continue;
				for( int64_t i = 0; i < guard; i++ )
					if( h[ (size_t)i ] != wh_context::GUARD_BYTE )
					{
						fprintf( stderr, "WH_GUARD_VIOLATION: wh_buffer_alloc buffer (%lld bytes): write at offset %lld\n", (long long)g.bytes,
							(long long)( side ? g.bytes + i : i - guard ) );
						break;
					}
			}
			dev = g.base;
		}
	}
	WH_HIP( hipFree( dev ) );
	return 0;
}

int wh_buffer_upload( wh_context* c, void* dev, const void* host, int64_t bytes )
{
	if( !c || !dev || !host || bytes < 0 ) { setError( "buffer_upload: bad argument" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	WH_HIP( hipMemcpyAsync( dev, host, (size_t)bytes, hipMemcpyHostToDevice, c->stream ) );
	WH_HIP( hipStreamSynchronize( c->stream ) );
	return 0;
}

int wh_buffer_upload_async( wh_context* c, void* dev, const void* host, int64_t bytes )
{
	if( !c || !dev || !host || bytes < 0 ) { setError( "buffer_upload_async: bad argument" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	WH_HIP( hipMemcpyAsync( dev, host, (size_t)bytes, hipMemcpyHostToDevice, c->stream ) );
	return 0;
}

int wh_buffer_download( wh_context* c, void* host, const void* dev, int64_t bytes )
{
	if( !c || !dev || !host || bytes < 0 ) { setError( "buffer_download: bad argument" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	WH_HIP( hipMemcpyAsync( host, dev, (size_t)bytes, hipMemcpyDeviceToHost, c->stream ) );
	WH_HIP( hipStreamSynchronize( c->stream ) );
	return 0;
}

int wh_context_synchronize( wh_context* c )
{
	if( !c ) return WH_E_INVALIDARG;
	WH_BIND( c->m );
	WH_HIP( hipStreamSynchronize( c->stream ) );
	return 0;
}

int wh_context_memory( const wh_context* c, int64_t* vramBytes )
{
	if( !c || !vramBytes ) return WH_E_INVALIDARG;
	*vramBytes = c->vram;
	return 0;
}

int wh_mel_spectrogram( wh_context* c, const float* pcmDev, int64_t nSamples, float* melDev, int64_t* nLenOut )
{
	if( !c || nSamples < 0 ) { setError( "mel: bad argument" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	const int64_t nLen = nSamples / 160;
	if( nLenOut ) *nLenOut = nLen;
	if( nLen == 0 ) return 0;	// less than one hop of audio: an empty spectrogram, like the reference (whisper.cpp:2080)
	if( !pcmDev || !melDev ) { setError( "mel: null buffer" ); return WH_E_INVALIDARG; }
	const wh_model* m = c->m;
	return profiled( c, KC_MEL, 2.0 * 2.0 * 400.0 * 201.0 * nLen, 4.0 * nSamples + 4.0 * 2.0 * nLen * m->hp.n_mels,
		[ & ]() { return launchMel( pcmDev, nSamples, m->at<float>( m->L.filters ), m->at<double>( m->L.dft ), melDev, nLen, m->hp.n_mels, c->melScratch, c->stream ); } );
}

int wh_mel_spectrogram_batch( wh_context* c, const float* pcmDev, int64_t nSamples, int64_t pcmStride, int batch, float* melDev, int64_t melStride )
{
	if( !c || nSamples < 0 || batch < 0 || pcmStride < 0 || melStride < 0 ) { setError( "mel_batch: bad argument" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	const int64_t nLen = nSamples / 160;
	if( nLen == 0 || batch == 0 ) return 0;
	if( !pcmDev || !melDev ) { setError( "mel_batch: null buffer" ); return WH_E_INVALIDARG; }
	const wh_model* m = c->m;
	if( melStride < nLen * m->hp.n_mels || pcmStride < nSamples ) { setError( "mel_batch: buffers overlap" ); return WH_E_INVALIDARG; }
	for( int b0 = 0; b0 < batch; b0 += MEL_BATCH_MAX )
	{
		const int nb = batch - b0 < MEL_BATCH_MAX ? batch - b0 : MEL_BATCH_MAX;
		const float* const pcm = pcmDev + (int64_t)b0 * pcmStride;
		float* const mel = melDev + (int64_t)b0 * melStride;
		int covered = 0;
		const int rc = profiled( c, KC_MEL, 2.0 * 2.0 * 400.0 * 201.0 * nLen * nb, ( 4.0 * nSamples + 4.0 * 2.0 * nLen * m->hp.n_mels ) * nb,
			[ & ]() {
				covered = launchMelBatch( pcm, nSamples, pcmStride, nb, m->at<float>( m->L.filters ), m->at<double>( m->L.dft ), mel, melStride, nLen, m->hp.n_mels, c->melScratch + 16, c->stream );
				return covered == 1 ? 0 : covered; } );
		if( rc ) return rc;
		if( covered == 1 )
			for( int b = 0; b < nb; b++ )
			{
				const int r1 = wh_mel_spectrogram( c, pcm + (int64_t)b * pcmStride, nSamples, mel + (int64_t)b * melStride, nullptr );
				if( r1 ) return r1;
			}
	}
	return 0;
}

int wh_mel_spectrogram_window( wh_context* c, const float* pcmDev, int64_t nSamples, int64_t frame0, int64_t nFrames, int64_t nChunks,
	int reusePreviousMax, float* melDev )
{
	if( !c || nSamples < 0 || frame0 < 0 || nFrames < 0 ) { setError( "mel_window: bad argument" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	if( nFrames == 0 ) return 0;
	if( !pcmDev || !melDev ) { setError( "mel_window: null buffer" ); return WH_E_INVALIDARG; }
	const wh_model* m = c->m;
	// frame f of the stream starts at sample f * 160; frames at or beyond the reader's chunk count are zero before normalisation
	const int64_t first = frame0 * 160;
	const int64_t remaining = nSamples > first ? nSamples - first : 0;
	int64_t valid = nChunks - frame0;
	valid = valid < 0 ? 0 : ( valid > nFrames ? nFrames : valid );
	return profiled( c, KC_MEL, 2.0 * 2.0 * 400.0 * 201.0 * nFrames, 4.0 * 160.0 * nFrames + 4.0 * 2.0 * nFrames * m->hp.n_mels,
		[ & ]() { return launchMelWindow( pcmDev + ( remaining > 0 ? first : 0 ), remaining, m->at<float>( m->L.filters ), m->at<double>( m->L.dft ), melDev,
			nFrames, valid, m->hp.n_mels, reusePreviousMax, c->melScratch, c->stream ); } );
}

// ------------------------------------------------------------------------------------------------------------------
// encoder
// ------------------------------------------------------------------------------------------------------------------
static GemmArgs plainGemm( const f16* A, const f16* W, int M, int N, int K )
{
	GemmArgs g;
	memset( &g, 0, sizeof( g ) );
	g.A = A; g.W = W; g.M = M; g.N = N; g.K = K;
	g.lda = K; g.Mb = M; g.aBatchStride = 0;
	g.ldc = N; g.cBatchStride = 0;
	g.scale = 1.0f;
	return g;
}

static int encodeImpl( wh_context* c, const float* melDev, int batch, int64_t melLen, int64_t melStride, const int32_t* melOffsets, const wh_mel_window* wins = nullptr );

int wh_encode_windows( wh_context* c, const wh_mel_window* windows, int batch )
{
	if( !windows ) { setError( "encode_windows: windows is null" ); return WH_E_INVALIDARG; }
	return encodeImpl( c, nullptr, batch, 0, 0, nullptr, windows );
}

int wh_encode( wh_context* c, const float* melDev, int batch, int64_t melLen, int64_t melStride, const int32_t* melOffsets )
{
	if( !c || !c->encStream || c->prof.on ) return encodeImpl( c, melDev, batch, melLen, melStride, melOffsets );
	// everything queued so far (PCM upload, spectrogram) -> encoder on the low-priority stream -> the decode stream waits for it
	WH_BIND( c->m );
	WH_HIP( hipEventRecord( c->encReady, c->stream ) );
	WH_HIP( hipStreamWaitEvent( c->encStream, c->encReady, 0 ) );
	hipStream_t const main = c->stream;
	c->stream = c->encStream;
	const int rc = encodeImpl( c, melDev, batch, melLen, melStride, melOffsets );
	c->stream = main;
	WH_CHECK( rc );
	WH_HIP( hipEventRecord( c->encDone, c->encStream ) );
	WH_HIP( hipStreamWaitEvent( c->stream, c->encDone, 0 ) );
	return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// WH_FLAG_PARITY_EXACT: the reference's graphs (whisper.cpp:1084-1496, :1508-1872) over the exact-order kernels of exact.hip
// ------------------------------------------------------------------------------------------------------------------
static int exactAlloc( wh_context* c, void** p, int64_t bytes )
{
	void* v = nullptr;
	WH_HIP( hipMalloc( &v, (size_t)bytes ) );
	WH_HIP( hipMemsetAsync( v, 0, (size_t)bytes, c->stream ) );
	c->ex.owned.push_back( v );
	c->vram += bytes;
	*p = v;
	return 0;
}
static int exactTables( wh_context* c )
{
	wh_context::Exact& e = c->ex;
	if( e.gelu ) return 0;
	std::vector<uint16_t> g( 65536 ), x( 65536 );
	exactBuildTables( g.data(), x.data() );
	WH_CHECK( exactAlloc( c, (void**)&e.gelu, 65536 * 2 ) );
	WH_CHECK( exactAlloc( c, (void**)&e.expt, 65536 * 2 ) );
	WH_HIP( hipMemcpyAsync( e.gelu, g.data(), 65536 * 2, hipMemcpyHostToDevice, c->stream ) );
	WH_HIP( hipMemcpyAsync( e.expt, x.data(), 65536 * 2, hipMemcpyHostToDevice, c->stream ) );
	WH_HIP( hipStreamSynchronize( c->stream ) );	   // the staging vectors go out of scope
	return 0;
}

static int encodeExact( wh_context* c, const float* melDev, int batch, int64_t melLen, int64_t melStride, const int32_t* melOffsets, const wh_mel_window* wins )
{
	const wh_model* m = c->m;
	const wh_hparams& hp = m->hp;
	const Layout& L = m->L;
	hipStream_t st = c->stream;
	const int d = hp.n_audio_state, H = hp.n_audio_head, T = c->T;
	wh_context::Exact& e = c->ex;
	WH_CHECK( exactTables( c ) );
	const int chunk = std::min( { wh_context::Exact::CHUNK, c->maxBatch, c->encChunk } );
	if( !e.x )
	{
		const int64_t rows = (int64_t)chunk * T;
		for( float** p : { &e.x, &e.cur, &e.q, &e.k, &e.v, &e.kqv } ) WH_CHECK( exactAlloc( c, (void**)p, rows * d * 4 ) );
		WH_CHECK( exactAlloc( c, (void**)&e.h, rows * 4 * d * 4 ) );
		WH_CHECK( exactAlloc( c, (void**)&e.conv1, rows * 2 * d * 4 ) );
		e.encWindows = chunk;
	}
	const float kScale = (float)pow( (double)( (float)d / (float)H ), -0.25 );
	auto mm = [ & ]( int64_t wOff, int N, int K, const float* X, int M, float* out, int64_t biasOff, bool useScale, bool gelu, const float* res ) -> int
	{
		return launchExactMulMat( m->at<f16>( wOff ), N, K, X, K, M, out, N, biasOff >= 0 ? m->at<float>( biasOff ) : nullptr, kScale, useScale,
			gelu ? e.gelu : nullptr, res, N, st );
	};
	for( int b0 = 0; b0 < batch; b0 += chunk )
	{
		const int nb = std::min( chunk, batch - b0 );
		const int M = nb * T;
		// the product's own conv input: fp16( mel ) time-major with the padding rows, exactly the operand ggml_conv_1d_1s builds (ggml.c:5270-5282)
		if( wins )
		{
			MelWindow* const stage = (MelWindow*)c->pinned;
			for( int i = 0; i < nb; i++ ) stage[ i ] = MelWindow{ wins[ b0 + i ].melDev, (long long)wins[ b0 + i ].melLen, wins[ b0 + i ].offset, 0 };
			WH_HIP( hipMemcpyAsync( c->melWindowsDev, stage, sizeof( MelWindow ) * nb, hipMemcpyHostToDevice, st ) );
		}
		else
		{
			for( int i = 0; i < nb; i++ ) c->pinned[ i ] = melOffsets ? melOffsets[ b0 + i ] : 0;
			WH_HIP( hipMemcpyAsync( c->melOffsetsDev, c->pinned, sizeof( int32_t ) * nb, hipMemcpyHostToDevice, st ) );
		}
		WH_CHECK( launchMelToConvInput( melDev ? melDev + (int64_t)b0 * melStride : nullptr, melStride, melLen, c->melOffsetsDev, wins ? c->melWindowsDev : nullptr,
			c->convIn, c->convInStride, hp.n_mels, 2 * T, nb, st ) );
		WH_HIP( hipStreamSynchronize( st ) );	   // the pinned staging is rewritten by the next chunk
		WH_CHECK( launchExactConv( m->at<f16>( L.conv1w ), conv1Kpad( hp ), hp.n_mels, c->convIn, true, c->convInStride, 2 * T, 1, m->at<float>( L.conv1b ), e.gelu,
			nullptr, e.conv1, 2ll * T * d, d, nb, st ) );
		WH_CHECK( launchExactConv( m->at<f16>( L.conv2w ), 3 * d, d, e.conv1, false, 2ll * T * d, 2 * T, 2, m->at<float>( L.conv2b ), e.gelu,
			m->at<float>( L.encPe ), e.x, (int64_t)T * d, d, nb, st ) );
		const int encLayers = g_opt.exactEncLayers >= 0 ? std::min( g_opt.exactEncLayers, hp.n_audio_layer ) : hp.n_audio_layer;
		for( int il = 0; il < encLayers; il++ )
		{
			const EncLayer& el = L.enc[ il ];
			WH_CHECK( launchExactNorm( e.x, m->at<float>( el.ln1w ), m->at<float>( el.ln1b ), e.cur, M, d, st ) );
			WH_CHECK( mm( el.wqkv, d, d, e.cur, M, e.q, el.bqkv, false, false, nullptr ) );
			WH_CHECK( mm( el.wqkv + 2ll * d * d, d, d, e.cur, M, e.k, -1, false, false, nullptr ) );
			WH_CHECK( mm( el.wqkv + 4ll * d * d, d, d, e.cur, M, e.v, el.bqkv + 8ll * d, false, false, nullptr ) );
			WH_CHECK( launchExactFlashAttn( e.q, e.k, e.v, e.kqv, nb, H, T, e.expt, st ) );
			WH_CHECK( mm( el.wo, d, d, e.kqv, M, e.x, el.bo, false, false, e.x ) );
			WH_CHECK( launchExactNorm( e.x, m->at<float>( el.ln2w ), m->at<float>( el.ln2b ), e.cur, M, d, st ) );
			WH_CHECK( mm( el.w1, 4 * d, d, e.cur, M, e.h, el.b1, false, true, nullptr ) );
			WH_CHECK( mm( el.w2, d, 4 * d, e.h, M, e.x, el.b2, false, false, e.x ) );
		}
		if( g_opt.exactEncLayers >= 0 ) continue;	   // debugging: the buffers hold the state after `encLayers` layers
		WH_CHECK( launchExactNorm( e.x, m->at<float>( L.lnPostW ), m->at<float>( L.lnPostB ), e.cur, M, d, st ) );
		for( int il = 0; il < hp.n_text_layer; il++ )
		{
			// Kcross = scale( mul_mat ), Vcross = mul_mat + bias, both copied into the FP16 caches (whisper.cpp:1448-1487)
			WH_CHECK( mm( L.wcross + 2ll * ( 2ll * il ) * d * d, d, d, e.cur, M, e.k, -1, true, false, nullptr ) );
			WH_CHECK( mm( L.wcross + 2ll * ( 2ll * il + 1 ) * d * d, d, d, e.cur, M, e.v, L.bcross + 4ll * ( 2ll * il + 1 ) * d, false, false, nullptr ) );
			const int64_t layerOff = ( (int64_t)il * c->maxBatch + b0 ) * T * d;
			WH_CHECK( launchExactPackHeads( e.k, c->crossK + layerOff, nb, T, T, 0, H, st ) );
			WH_CHECK( launchExactPackHeads( e.v, c->crossV + layerOff, nb, T, T, 0, H, st ) );
		}
	}
	c->encoded = true;
	c->lastEncBatch = batch;
	c->lastBatch = batch * c->hyp;
	return 0;
}

// whisper_decode: every sequence's nTokens tokens at nPast; logits and probabilities of the LAST token of every sequence into c->logits / c->probs
static int decodeExact( wh_context* c, int batch, int nTokens, int nPast )
{
	const wh_model* m = c->m;
	const wh_hparams& hp = m->hp;
	const Layout& L = m->L;
	hipStream_t st = c->stream;
	const int d = hp.n_text_state, H = hp.n_text_head, T = c->T, N = nTokens;
	wh_context::Exact& e = c->ex;
	WH_CHECK( exactTables( c ) );
	const int64_t rows = (int64_t)batch * N;
	if( rows > e.decRows )
	{
		WH_HIP( hipStreamSynchronize( st ) );
		for( float** p : { &e.dx, &e.dcur, &e.dq, &e.dk, &e.dv, &e.dkqv } ) WH_CHECK( exactAlloc( c, (void**)p, rows * d * 4 ) );
		WH_CHECK( exactAlloc( c, (void**)&e.dh, rows * 4 * d * 4 ) );
		e.decRows = rows;
	}
	const int maxKeys = std::max( T, nPast + N );
	const int64_t needScores = (int64_t)batch * H * N * maxKeys;
	if( needScores > e.scoreFloats )
	{
		WH_HIP( hipStreamSynchronize( st ) );
		WH_CHECK( exactAlloc( c, (void**)&e.scores, needScores * 4 ) );
		e.scoreFloats = needScores;
	}
	const float s = (float)pow( (double)( (float)d / (float)H ), -0.25 );
	const int M = (int)rows;
	auto mm = [ & ]( int64_t wOff, int Nn, int K, const float* X, float* out, int64_t biasOff, bool useScale, bool gelu, const float* res ) -> int
	{
		return launchExactMulMat( m->at<f16>( wOff ), Nn, K, X, K, M, out, Nn, biasOff >= 0 ? m->at<float>( biasOff ) : nullptr, s, useScale, gelu ? e.gelu : nullptr, res, Nn, st );
	};
	// token + positional embedding: one FP32 add per element, which launchEmbed already is (ggml_add of get_rows, whisper.cpp:1560-1571)
	WH_CHECK( launchEmbed( c->tokensDev, m->at<f16>( L.te ), m->at<float>( L.decPe ), e.dx, M, N, nPast, nullptr, d, hp.n_vocab, hp.n_text_ctx, st ) );
	for( int il = 0; il < hp.n_text_layer; il++ )
	{
		const DecLayer& dl = L.dec[ il ];
		WH_CHECK( launchExactNorm( e.dx, m->at<float>( dl.ln1w ), m->at<float>( dl.ln1b ), e.dcur, M, d, st ) );
		WH_CHECK( mm( dl.wqkv, d, d, e.dcur, e.dq, dl.bqkv, true, false, nullptr ) );						// Qcur = scale( mul_mat + b )
		WH_CHECK( mm( dl.wqkv + 2ll * d * d, d, d, e.dcur, e.dk, -1, true, false, nullptr ) );				// Kcur = scale( mul_mat )
		WH_CHECK( mm( dl.wqkv + 4ll * d * d, d, d, e.dcur, e.dv, dl.bqkv + 8ll * d, false, false, nullptr ) );	// Vcur = mul_mat + b
		f16* const sk = c->selfK + (int64_t)il * c->maxSeq * hp.n_text_ctx * d;
		f16* const sv = c->selfV + (int64_t)il * c->maxSeq * hp.n_text_ctx * d;
		WH_CHECK( launchExactPackHeads( e.dk, sk, batch, N, hp.n_text_ctx, nPast, H, st ) );
		WH_CHECK( launchExactPackHeads( e.dv, sv, batch, N, hp.n_text_ctx, nPast, H, st ) );
		WH_CHECK( launchExactDecAttention( e.dq, sk, sv, e.scores, e.dkqv, batch, N, nPast + N, H, hp.n_text_ctx, 1, nPast, true, c->parityThreads, e.expt, st ) );
		WH_CHECK( mm( dl.wo, d, d, e.dkqv, e.dx, dl.bo, false, false, e.dx ) );
		WH_CHECK( launchExactNorm( e.dx, m->at<float>( dl.lncw ), m->at<float>( dl.lncb ), e.dcur, M, d, st ) );
		WH_CHECK( mm( dl.wcq, d, d, e.dcur, e.dq, dl.bcq, true, false, nullptr ) );
		const int64_t crossOff = (int64_t)il * c->maxBatch * T * d;
		WH_CHECK( launchExactDecAttention( e.dq, c->crossK + crossOff, c->crossV + crossOff, e.scores, e.dkqv, batch, N, T, H, T, c->hyp, 0, false, c->parityThreads, e.expt, st ) );
		WH_CHECK( mm( dl.wco, d, d, e.dkqv, e.dx, dl.bco, false, false, e.dx ) );
		WH_CHECK( launchExactNorm( e.dx, m->at<float>( dl.ln2w ), m->at<float>( dl.ln2b ), e.dcur, M, d, st ) );
		WH_CHECK( mm( dl.w1, 4 * d, d, e.dcur, e.dh, dl.b1, false, true, nullptr ) );
		WH_CHECK( mm( dl.w2, d, 4 * d, e.dh, e.dx, dl.b2, false, false, e.dx ) );
	}
	WH_CHECK( launchExactNorm( e.dx, m->at<float>( L.decLnW ), m->at<float>( L.decLnB ), e.dcur, M, d, st ) );
	// logits of the last token of every sequence: row b of the product is row b * N + N - 1 of the normalised stream
	WH_CHECK( launchExactMulMat( m->at<f16>( L.te ), hp.n_vocab, d, e.dcur + (int64_t)( N - 1 ) * d, (int64_t)N * d, batch, c->logits, hp.n_vocab, nullptr, 0.0f, false, nullptr,
		nullptr, 0, st ) );
	WH_CHECK( launchExactSoftMax( c->logits, c->probs, batch, hp.n_vocab, e.expt, st ) );
	return 0;
}

static int encodeImpl( wh_context* c, const float* melDev, int batch, int64_t melLen, int64_t melStride, const int32_t* melOffsets, const wh_mel_window* wins )
{
	if( !c || ( !melDev && !wins ) || batch <= 0 || batch > c->maxBatch || ( !wins && melLen <= 0 ) ) { setError( "encode: bad argument" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	const wh_model* m = c->m;
	const wh_hparams& hp = m->hp;
	const Layout& L = m->L;
	hipStream_t st = c->stream;
	const int d = hp.n_audio_state, H = hp.n_audio_head, T = c->T;
	const int batchAll = batch;

	if( batch > wh_context::PIN_WINDOWS ) { setError( "encode: batch too large" ); return WH_E_INVALIDARG; }
	if( c->flags & WH_FLAG_PARITY_EXACT ) return encodeExact( c, melDev, batch, melLen, melStride, melOffsets, wins );
	if( batch > c->encChunk && ( c->flags & WH_FLAG_DEBUG_CAPTURE ) ) { setError( "encode: the probe-point capture needs a batch of one encoder chunk" ); return WH_E_INVALIDARG; }
	const bool gated = ( g_tuning & TUNE_ENC_SERIAL ) && batch >= ENC_SERIAL_MIN_WINDOWS && liveContexts( m ).load( std::memory_order_relaxed ) > 1;
	if( gated )
	{
		std::lock_guard<std::mutex> lk( g_encGateMx );
		EncGate& gate = g_encGate[ m->device & 63 ];
		if( gate.last && gate.owner != c ) WH_HIP( hipStreamWaitEvent( st, gate.last, 0 ) );
	}
	// offsets go through pinned staging (ints [0, 4096)): the copy is truly asynchronous and the call never blocks.
	// The staging is rewritten by the next wh_encode only, which the stream orders after this copy has been consumed
	// as long as the caller synchronises once per window (wh_decode / wh_decode_window_finish do).
	const MelWindow* winsDev = nullptr;
	if( wins )
	{
		// per-window sources (wh_encode_windows): descriptors through the same staging, 6 ints each
		static_assert( sizeof( MelWindow ) == 24 && sizeof( wh_mel_window ) == 24, "window descriptor layout" );
		if( (size_t)batch * sizeof( MelWindow ) > wh_context::PIN_WINDOWS * sizeof( int32_t ) ) { setError( "encode_windows: batch too large" ); return WH_E_INVALIDARG; }
		MelWindow* const stage = (MelWindow*)c->pinned;
		for( int i = 0; i < batch; i++ )
		{
			if( wins[ i ].melDev && ( wins[ i ].melLen <= 0 || wins[ i ].offset < 0 ) ) { setError( "encode_windows: bad window" ); return WH_E_INVALIDARG; }
			stage[ i ] = MelWindow{ wins[ i ].melDev, (long long)wins[ i ].melLen, wins[ i ].offset, 0 };
		}
		WH_HIP( hipMemcpyAsync( c->melWindowsDev, stage, sizeof( MelWindow ) * batch, hipMemcpyHostToDevice, st ) );
		winsDev = c->melWindowsDev;
	}
	else
	{
		for( int i = 0; i < batch; i++ ) c->pinned[ i ] = melOffsets ? melOffsets[ i ] : 0;
		WH_HIP( hipMemcpyAsync( c->melOffsetsDev, c->pinned, sizeof( int32_t ) * batch, hipMemcpyHostToDevice, st ) );
	}
	// A batch larger than the encoder's chunk is encoded chunk by chunk through the same activations: windows are independent, the
	// products are MFMA-bound and saturated at a chunk's row count, and only the cross-attention caches (written in place at the
	// chunk's window offset) are sized for the whole batch.
	const int nChunks = ( batchAll + c->encChunk - 1 ) / c->encChunk;
	const int perChunk = ( batchAll + nChunks - 1 ) / nChunks;
	for( int b0 = 0; b0 < batchAll; b0 += perChunk )
	{
	batch = std::min( perChunk, batchAll - b0 );
	const int M = batch * T;
	WH_CHECK( profiled( c, KC_MEL_TO_CONV, 0.0, 6.0 * batch * 2.0 * T * hp.n_mels,
		[ & ]() { return launchMelToConvInput( melDev ? melDev + (int64_t)b0 * melStride : nullptr, melStride, melLen, c->melOffsetsDev + b0, winsDev ? winsDev + b0 : nullptr,
			c->convIn, c->convInStride, hp.n_mels, 2 * T, batch, st ); } ) );

	// conv1 (k=3, stride 1, pad 1) + bias + GELU as an implicit GEMM over the padded time-major input:
	// row t of the im2col matrix is the contiguous slice starting at padded row t (whisper.cpp:1127-1136; ggml.c:5199-5318)
	{
		GemmArgs g = plainGemm( c->convIn, m->at<f16>( L.conv1w ), batch * 2 * T, d, conv1Kpad( hp ) );
		g.lda = hp.n_mels; g.Mb = 2 * T; g.aBatchStride = c->convInStride;
		g.epi = EPI_F16_GELU;
		g.bias = m->at<float>( L.conv1b );
		g.out16 = c->conv1Out + d;	 // padded row t+1
		g.ldc = d; g.cBatchStride = c->conv1Stride;
		WH_CHECK( gemmP( c, g, false ) );
	}
	WH_CHECK( capture( c, c->capTemp1, c->conv1Out, (int64_t)batch * c->conv1Stride, (int64_t)c->maxBatch * c->conv1Stride ) );	// "enc.temp1"
	// conv2 (stride 2) + bias + GELU + positional embedding -> residual stream x [batch*T][d] (whisper.cpp:1138-1167)
	{
		GemmArgs g = plainGemm( c->conv1Out, m->at<f16>( L.conv2w ), M, d, 3 * d );
		g.lda = 2 * d; g.Mb = T; g.aBatchStride = c->conv1Stride;
		g.epi = EPI_CONV2;
		g.bias = m->at<float>( L.conv2b );
		g.pe = m->at<float>( L.encPe );
		g.out32 = c->x; g.ldc = d;
		WH_CHECK( gemmP( c, g, false ) );
	}
	WH_CHECK( capture( c, c->capLayer0In, c->x, (int64_t)M * d, (int64_t)c->maxBatch * T * d ) );	// "enc.layer[ 0 ].in"
	for( int il = 0; il < hp.n_audio_layer; il++ )
	{
		const EncLayer& e = L.enc[ il ];
		WH_CHECK( lnP( c, c->x, m->at<float>( e.ln1w ), m->at<float>( e.ln1b ), c->xn, M, d ) );
		{
			GemmArgs g = plainGemm( c->xn, m->at<f16>( e.wqkv ), M, 3 * d, d );
			g.epi = EPI_QKV_ENC;
			g.bias = m->at<float>( e.bqkv );
			g.q = c->q; g.k = c->k; g.v = c->vT;
			g.T = T; g.Tpad = c->Tpad; g.H = H; g.B = batch;
			WH_CHECK( gemmP( c, g, false ) );
		}
		WH_CHECK( profiled( c, KC_ATTN_ENC, 4.0 * batch * H * (double)T * T * HEAD_DIM, 2.0 * 4.0 * batch * H * (double)T * HEAD_DIM,
			[ & ]() { return launchAttentionEnc( c->q, c->k, c->vT, c->attn, batch, H, T, c->Tpad, ( c->flags & WH_FLAG_PARITY_PV ) != 0, m->at<f16>( L.expTab ), st ); } ) );
		if( il == 0 ) WH_CHECK( capture( c, c->capEncKqv, c->attn, (int64_t)M * d, (int64_t)c->maxBatch * T * d ) );	// "enc-KQV"
		{
			GemmArgs g = plainGemm( c->attn, m->at<f16>( e.wo ), M, d, d );
			g.epi = EPI_F32;
			g.bias = m->at<float>( e.bo );
			g.res = c->x; g.out32 = c->x;
			WH_CHECK( gemmP( c, g, false ) );
		}
		WH_CHECK( lnP( c, c->x, m->at<float>( e.ln2w ), m->at<float>( e.ln2b ), c->xn, M, d ) );
		{
			GemmArgs g = plainGemm( c->xn, m->at<f16>( e.w1 ), M, 4 * d, d );
			g.epi = EPI_F16_GELU;
			g.bias = m->at<float>( e.b1 );
			g.out16 = c->h;
			WH_CHECK( gemmP( c, g, false ) );
		}
		{
			GemmArgs g = plainGemm( c->h, m->at<f16>( e.w2 ), M, d, 4 * d );
			g.epi = EPI_F32;
			g.bias = m->at<float>( e.b2 );
			g.res = c->x; g.out32 = c->x;
			WH_CHECK( gemmP( c, g, false ) );
		}
	}
	WH_CHECK( lnP( c, c->x, m->at<float>( L.lnPostW ), m->at<float>( L.lnPostB ), c->xn, M, d ) );
	// cross-attention K/V of every decoder layer in one product (whisper.cpp:1448-1487)
	{
		GemmArgs g = plainGemm( c->xn, m->at<f16>( L.wcross ), M, 2 * hp.n_text_layer * d, d );
		g.epi = EPI_CROSS_KV;
		g.bias = m->at<float>( L.bcross );
		g.scale = (float)pow( (double)( (float)d / (float)H ), -0.25 );
		// [layer][window][head][T][64]: the chunk's first window; the layer stride stays maxBatch windows
		g.k = c->crossK + (int64_t)b0 * T * d; g.v = c->crossV + (int64_t)b0 * T * d;
		g.T = T; g.H = H; g.B = c->maxBatch;
		WH_CHECK( gemmP( c, g, false ) );
	}
	}	// chunks
	batch = batchAll;
	if( gated )
	{
		std::lock_guard<std::mutex> lk( g_encGateMx );
		if( !c->encGateEv ) WH_HIP( hipEventCreateWithFlags( &c->encGateEv, hipEventDisableTiming ) );
		WH_HIP( hipEventRecord( c->encGateEv, st ) );
		EncGate& gate = g_encGate[ m->device & 63 ];
		gate.last = c->encGateEv;
		gate.owner = c;
	}
	c->encoded = true;
	c->lastEncBatch = batch;
	c->lastBatch = batch * c->hyp;
	return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// decoder
// ------------------------------------------------------------------------------------------------------------------
// The decoder graph up to the logits of the last token of every sequence. With devState the position comes from device
// memory (c->seqPos, one per sequence), which makes the launch sequence identical for every token: that is what gets captured
// into a hipGraph by wh_decode_greedy. Single-token steps of up to 16 sequences take the gemv path (weights streamed
// once, all loads of a wave in flight, LayerNorm fused into the product that consumes it); anything larger (prompt
// steps) takes the M <= 32 skinny or the tiled kernel.
static int decodeGraph( wh_context* c, int batch, int nTokens, int nPast, bool devState )
{
	if( c->flags & WH_FLAG_PARITY_EXACT ) { setError( "WH_FLAG_PARITY_EXACT: host-stepped decoding only (wh_decode + wh_sample_best)" ); return WH_E_INVALIDARG; }
	const wh_model* m = c->m;
	const wh_hparams& hp = m->hp;
	const Layout& L = m->L;
	hipStream_t st = c->stream;
	const int d = hp.n_text_state, H = hp.n_text_head;
	const int M = batch * nTokens;
	const float kqScale = (float)pow( (double)( (float)d / (float)H ), -0.25 );
	const int parity = ( c->flags & WH_FLAG_PARITY_PV ) ? c->parityThreads : 0;
	const int* const nPastDev = devState ? c->seqPos : nullptr;
	// the weight-streaming kernels (gemvFused up to 128 rows, gemmDecRows up to 512) are sized and tuned for single-token steps of a lock-step batch; a PROMPT
	// step of 129 .. 512 rows (one stream with a carried-over prompt of ~224 tokens, small batches with a prompt) keeps the M-tiled MFMA kernel it always had
	// (raising GEMV_MAX_ROWS to 512 had silently re-routed those, summation order included)
	const bool gemv = M <= ( nTokens == 1 ? GEMV_MAX_ROWS : GEMV_FUSED_MAX_ROWS ) && ( d % 128 ) == 0;
	const bool fuseLn = gemv && d <= 1280 && M <= 32 && ( M <= 16 || ( g_tuning & TUNE_GEMV_LN_BLOCK ) || !( g_tuning & TUNE_LN_SEPARATE_BIGM ) );
	// decode steps: LayerNorm + this head's Q/K/V rows + cache append + self-attention in one launch
	// (from 9 sequences up: with fewer, one workgroup per (head, sequence) leaves the 6 d^2 bytes of QKV weights to H CUs at
	// ~25 GB/s each -- 15.7 us per layer at one sequence -- and the separate gemv + attention launches are faster: 74.3 vs 86.6 ms
	// per window at batch 1, 100.4 vs 101.9 at 7)
	const bool fuseSelf = nTokens == 1 && d <= 1280 && hp.n_text_ctx <= 512 && parity == 0 && batch > 8 && batch <= g_opt.selfFuseMaxRows && ( g_tuning & TUNE_FUSE_SELF_BLOCK );
	// decode steps: the cross-attention kernel normalises the residual row and projects its own head's query
	const bool fuseCrossQ = nTokens == 1 && d <= 1280 && parity <= 8 && ( g_tuning & TUNE_FUSE_CROSS_Q );

	auto product = [ & ]( GemmArgs& g, const float* lnW, const float* lnB ) -> int
	{
		g.nPastDev = nPastDev;
		if( !gemv ) return gemmP( c, g, true, true );
		if( g.N <= d ) g.splitScratch = c->splitK;
		if( lnW && fuseLn )
		{
			g.lnX = c->dx; g.lnW = lnW; g.lnB = lnB;
		}
		const double flops = 2.0 * g.M * g.N * g.K;
		const double bytes = 2.0 * g.N * g.K + 2.0 * g.M * g.K + ( g.out32 ? 4.0 : 2.0 ) * g.M * g.N;
		return profiled( c, KC_GEMV, flops, bytes, [ & ]() { return launchGemv( g, st ); } );
	};

	WH_CHECK( profiled( c, KC_EMBED, 1.0 * M * d, 10.0 * M * d,
		[ & ]() { return launchEmbed( c->tokensDev, m->at<f16>( L.te ), m->at<float>( L.decPe ), c->dx, M, nTokens, nPast, nPastDev, d, hp.n_vocab, hp.n_text_ctx, st ); } ) );

	// ---- single-token steps of up to 4 sequences (one stream through iContext::runFull): decode1.hip. Eight launches per layer,
	// each of ~256 workgroups, each pulling the weights of the next one into the L2 of the XCD that will read them.
	const bool small = nTokens == 1 && batch <= SMALL_MAX_ROWS && c->hyp == 1 && parity == 0 && d <= 1280 && ( d % 64 ) == 0 && c->T <= 1536 &&
		!( c->flags & WH_FLAG_DEBUG_CAPTURE ) && ( g_tuning & TUNE_DECODE_SMALL );
	if( small )
	{
#ifdef WH_PROBES
		const bool pfOn = ( g_tuning & TUNE_DECODE_PREFETCH ) != 0;	   // measured slower: a probe build's switch only
#else
		const bool pfOn = false;
#endif
		// what a gemvSmall launch of N rows x K columns streams per workgroup (must mirror launchGemvSmall's row split)
		auto gemvChunk = [ & ]( int N, int K ) -> int
		{
			const int nch = ( K + 511 ) >> 9;
			int rw = ( N + 1023 ) / 1024;
			if( rw < 1 ) rw = 1;
			const int mr = batch <= 2 ? batch : 4;
			while( rw > 1 && ( rw * nch > 16 || rw * 4 * mr > 64 ) ) rw--;
			return rw * 4 * K * 2;
		};
		auto hint = [ & ]( const void* p, int64_t bytes, int chunk ) -> PrefetchHint
		{
			PrefetchHint h = { nullptr, 0, 0 };
			if( pfOn ) { h.ptr = p; h.bytes = bytes; h.chunkBytes = chunk; }
			return h;
		};
		auto smallGemv = [ & ]( SmallGemvArgs& a ) -> int
		{
			a.g.nPastDev = nPastDev;
			const GemmArgs& g = a.g;
			const double flops = 2.0 * g.M * g.N * g.K;
			const double bytes = 2.0 * g.N * g.K + 2.0 * g.M * g.K + ( g.out32 ? 4.0 : 2.0 ) * g.M * g.N;
			return profiled( c, KC_GEMV, flops, bytes, [ & ]() { return launchGemvSmall( a, st ); } );
		};
		const int64_t headKv = (int64_t)c->T * HEAD_DIM * 2;	   // bytes of one (sequence, head) of a cross-attention cache
		for( int il = 0; il < hp.n_text_layer; il++ )
		{
			const DecLayer& e = L.dec[ il ];
			const int64_t selfLayer = (int64_t)il * c->maxSeq * hp.n_text_ctx * d;
			const int64_t crossLayer = (int64_t)il * c->maxBatch * c->T * d;
			// 1. LayerNorm + Q/K/V (cache append)                          next but one: the self-attention output projection
			{
				SmallGemvArgs a = {};
				a.g = plainGemm( nullptr, m->at<f16>( e.wqkv ), M, 3 * d, d );
				a.g.epi = EPI_QKV_DEC; a.g.bias = m->at<float>( e.bqkv ); a.g.scale = kqScale;
				a.g.q = c->dq; a.g.k = c->selfK + selfLayer; a.g.v = c->selfV + selfLayer;
				a.g.H = H; a.g.nTok = 1; a.g.nPast = nPast; a.g.textCtx = hp.n_text_ctx;
				a.pro = 1; a.g.lnX = c->dx; a.g.lnW = m->at<float>( e.ln1w ); a.g.lnB = m->at<float>( e.ln1b );
				a.pf[ 0 ] = hint( m->at<f16>( e.wo ), (int64_t)d * d * 2, gemvChunk( d, d ) );
				WH_CHECK( smallGemv( a ) );
			}
			// 2. self-attention (a head per workgroup: at most n_text_ctx keys)
			{
				DecAttnArgs a = {};
				a.q = c->dq; a.kc = c->selfK + selfLayer; a.vc = c->selfV + selfLayer; a.out = c->dattn;
				a.batch = batch; a.H = H; a.nTok = 1; a.nKeys = nPast + 1; a.keyStride = hp.n_text_ctx;
				a.causal = 1; a.nPast = nPast; a.parityThreads = 0; a.nPastDev = nPastDev;
				if( devState ) a.nKeys = hp.n_text_ctx;
				WH_CHECK( attnDecP( c, a, devState ? c->profKeysHint : -1 ) );
				if( il == 0 && !devState )
				{
					WH_CHECK( capture( c, c->capDecKqvSelf, c->dattn, (int64_t)M * d, (int64_t)c->maxRows * d ) );	 // "dec-KQV" (self)
					c->capDecRows = M;
				}
			}
			// 3. output projection + residual                              next: the query rows and the keys of the cross-attention
			{
				SmallGemvArgs a = {};
				a.g = plainGemm( c->dattn, m->at<f16>( e.wo ), M, d, d );
				a.g.epi = EPI_F32; a.g.bias = m->at<float>( e.bo ); a.g.res = c->dx; a.g.out32 = c->dx;
				a.pf[ 0 ] = hint( m->at<f16>( e.wcq ), (int64_t)d * d * 2, HEAD_DIM * d * 2 );
				a.pf[ 1 ] = hint( c->crossK + crossLayer, (int64_t)batch * H * headKv, (int)headKv );
				WH_CHECK( smallGemv( a ) );
			}
			// 4. / 5. cross-attention over 8 key ranges per head            next: the values, then the output projection
			CrossSplitArgs x = {};
			x.lnX = c->dx; x.lnW = m->at<float>( e.lncw ); x.lnB = m->at<float>( e.lncb );
			x.qW = m->at<f16>( e.wcq ); x.qB = m->at<float>( e.bcq ); x.qScale = kqScale;
			x.kc = c->crossK + crossLayer; x.vc = c->crossV + crossLayer;
			x.scores = c->crossScores; x.splitMax = c->crossSplitMax; x.part = c->crossPart;
			x.batch = batch; x.H = H; x.nKeys = c->T; x.keyStride = c->T;
			{
				const double bytes = 2.0 * d * d + 2.0 * M * c->T * d + 4.0 * M * d;
				x.pf[ 0 ] = hint( c->crossV + crossLayer, (int64_t)batch * H * headKv, (int)headKv );
				WH_CHECK( profiled( c, KC_ATTN_DEC_CROSS, 2.0 * M * d * d + 2.0 * M * c->T * d, bytes, [ & ]() { return launchCrossScores( x, st ); } ) );
				x.pf[ 0 ] = hint( m->at<f16>( e.wco ), (int64_t)d * d * 2, gemvChunk( d, d ) );
				WH_CHECK( profiled( c, KC_ATTN_DEC_CROSS, 2.0 * M * c->T * d, 2.0 * M * c->T * d + 4.0 * M * H * c->T, [ & ]() { return launchCrossSoftmaxPV( x, st ); } ) );
			}
			// 6. splits combined + output projection + residual              next: the MLP up-projection
			{
				SmallGemvArgs a = {};
				a.g = plainGemm( nullptr, m->at<f16>( e.wco ), M, d, d );
				a.g.epi = EPI_F32; a.g.bias = m->at<float>( e.bco ); a.g.res = c->dx; a.g.out32 = c->dx;
				a.pro = 3; a.part = c->crossPart;
				a.pf[ 0 ] = hint( m->at<f16>( e.w1 ), (int64_t)4 * d * d * 2, gemvChunk( 4 * d, d ) );
				WH_CHECK( smallGemv( a ) );
			}
			// 7. LayerNorm + MLP up + GELU                                    next: the MLP down-projection
			{
				SmallGemvArgs a = {};
				a.g = plainGemm( nullptr, m->at<f16>( e.w1 ), M, 4 * d, d );
				a.g.epi = EPI_F16_GELU; a.g.bias = m->at<float>( e.b1 ); a.g.out16 = c->dh;
				a.pro = 1; a.g.lnX = c->dx; a.g.lnW = m->at<float>( e.ln2w ); a.g.lnB = m->at<float>( e.ln2b );
				a.pf[ 0 ] = hint( m->at<f16>( e.w2 ), (int64_t)4 * d * d * 2, gemvChunk( d, 4 * d ) );
				WH_CHECK( smallGemv( a ) );
			}
			// 8. MLP down + residual                                          next: the following layer's Q/K/V rows
			{
				SmallGemvArgs a = {};
				a.g = plainGemm( c->dh, m->at<f16>( e.w2 ), M, d, 4 * d );
				a.g.epi = EPI_F32; a.g.bias = m->at<float>( e.b2 ); a.g.res = c->dx; a.g.out32 = c->dx;
				if( il + 1 < hp.n_text_layer )
					a.pf[ 0 ] = hint( m->at<f16>( L.dec[ il + 1 ].wqkv ), (int64_t)3 * d * d * 2, gemvChunk( 3 * d, d ) );
				WH_CHECK( smallGemv( a ) );
			}
		}
		// final norm fused into the vocabulary projection's prologue (1621 workgroups normalise one row each: 4 KB from L2)
		{
			SmallGemvArgs a = {};
			a.g = plainGemm( nullptr, m->at<f16>( L.te ), batch, hp.n_vocab, d );
			a.g.epi = EPI_F32; a.g.out32 = c->logits; a.g.ldc = hp.n_vocab;
			a.pro = 1; a.g.lnX = c->dx; a.g.lnW = m->at<float>( L.decLnW ); a.g.lnB = m->at<float>( L.decLnB );
			WH_CHECK( smallGemv( a ) );
		}
		return 0;
	}

	for( int il = 0; il < hp.n_text_layer; il++ )
	{
		const DecLayer& e = L.dec[ il ];
		const int64_t selfLayer = (int64_t)il * c->maxSeq * hp.n_text_ctx * d;
		const int64_t crossLayer = (int64_t)il * c->maxBatch * c->T * d;
		// self-attention
		if( fuseSelf )
		{
			DecSelfArgs a = {};
			a.x = c->dx; a.lnW = m->at<float>( e.ln1w ); a.lnB = m->at<float>( e.ln1b );
			a.wqkv = m->at<f16>( e.wqkv ); a.bqkv = m->at<float>( e.bqkv ); a.scale = kqScale;
			a.kc = c->selfK + selfLayer; a.vc = c->selfV + selfLayer; a.out = c->dattn;
			a.batch = batch; a.H = H; a.keyStride = hp.n_text_ctx; a.nPast = nPast; a.nPastDev = nPastDev;
			const double keys = devState ? c->profKeysHint : nPast + 1;
			// algorithmic: the fused [3d][d] weight once, the residual rows, the cached K and V rows, the appended rows, the output
			const double bytes = 2.0 * 3.0 * d * d + 4.0 * M * d + 2.0 * 2.0 * M * ( keys - 1 ) * d + 2.0 * 2.0 * M * d + 2.0 * M * d;
			const double flops = 2.0 * M * 3.0 * d * d + 4.0 * M * keys * d;
			WH_CHECK( profiled( c, KC_SELF_BLOCK, flops, bytes, [ & ]() { return launchSelfBlockDec( a, st ); } ) );
			if( il == 0 && !devState )
			{
				WH_CHECK( capture( c, c->capDecKqvSelf, c->dattn, (int64_t)M * d, (int64_t)c->maxRows * d ) );	 // "dec-KQV" (self)
				c->capDecRows = M;
			}
		}
		else
		{
		if( !fuseLn ) WH_CHECK( lnP( c, c->dx, m->at<float>( e.ln1w ), m->at<float>( e.ln1b ), c->dxn, M, d, true ) );
		{
			GemmArgs g = plainGemm( c->dxn, m->at<f16>( e.wqkv ), M, 3 * d, d );
			g.epi = EPI_QKV_DEC;
			g.bias = m->at<float>( e.bqkv );
			g.scale = kqScale;
			g.q = c->dq; g.k = c->selfK + selfLayer; g.v = c->selfV + selfLayer;
			g.H = H; g.nTok = nTokens; g.nPast = nPast; g.textCtx = hp.n_text_ctx;
			WH_CHECK( product( g, m->at<float>( e.ln1w ), m->at<float>( e.ln1b ) ) );
		}
		{
			DecAttnArgs a = {};
			a.q = c->dq; a.kc = c->selfK + selfLayer; a.vc = c->selfV + selfLayer; a.out = c->dattn;
			a.batch = batch; a.H = H; a.nTok = nTokens; a.nKeys = nPast + nTokens; a.keyStride = hp.n_text_ctx;
			a.causal = 1; a.nPast = nPast; a.parityThreads = parity; a.nPastDev = nPastDev;
			if( devState ) a.nKeys = hp.n_text_ctx;	  // upper bound for the argument check; the kernel reads the real value
			WH_CHECK( attnDecP( c, a, devState ? c->profKeysHint : -1 ) );
			if( il == 0 && !devState )
			{
				WH_CHECK( capture( c, c->capDecKqvSelf, c->dattn, (int64_t)M * d, (int64_t)c->maxRows * d ) );	 // "dec-KQV" (self)
				c->capDecRows = M;
			}
		}
		}
		{
			GemmArgs g = plainGemm( c->dattn, m->at<f16>( e.wo ), M, d, d );
			g.epi = EPI_F32; g.bias = m->at<float>( e.bo ); g.res = c->dx; g.out32 = c->dx;
			WH_CHECK( product( g, nullptr, nullptr ) );
		}
		// cross-attention
		if( !fuseCrossQ )
		{
			if( !fuseLn ) WH_CHECK( lnP( c, c->dx, m->at<float>( e.lncw ), m->at<float>( e.lncb ), c->dxn, M, d, true ) );
			GemmArgs g = plainGemm( c->dxn, m->at<f16>( e.wcq ), M, d, d );
			g.epi = EPI_Q_DEC; g.bias = m->at<float>( e.bcq ); g.scale = kqScale; g.q = c->dq;
			WH_CHECK( product( g, m->at<float>( e.lncw ), m->at<float>( e.lncb ) ) );
		}
		{
			DecAttnArgs a = {};
			a.q = c->dq; a.kc = c->crossK + crossLayer; a.vc = c->crossV + crossLayer; a.out = c->dattn;
			a.batch = batch; a.H = H; a.nTok = nTokens; a.nKeys = c->T; a.keyStride = c->T;
			a.causal = 0; a.nPast = 0; a.parityThreads = parity; a.nPastDev = nullptr;
			a.group = c->hyp;
			// the rows of a multi-token (prompt) step see the same keys -- there is no causal mask across the encoder output --
			// so they are to the kernel what the hypotheses of a window are: one pass over the window's K/V for all of them
			// (as separate query tokens every token's workgroup re-read it: 2.4 GB instead of 0.69 GB per launch at 112 windows)
			{
				const int rowsPerWindow = c->hyp * nTokens;
				const bool groupable = rowsPerWindow <= 5 || rowsPerWindow == 8;
				if( nTokens > 1 && parity == 0 && groupable )
				{
					a.batch = batch * nTokens; a.nTok = 1; a.group = rowsPerWindow;
				}
			}
			if( fuseCrossQ )
			{
				a.lnX = c->dx; a.lnW = m->at<float>( e.lncw ); a.lnB = m->at<float>( e.lncb );
				a.qW = m->at<f16>( e.wcq ); a.qB = m->at<float>( e.bcq ); a.qScale = kqScale;
			}
			WH_CHECK( attnDecP( c, a ) );
			if( il == 0 && !devState ) WH_CHECK( capture( c, c->capDecKqvCross, c->dattn, (int64_t)M * d, (int64_t)c->maxRows * d ) );	  // "dec-KQV" (cross)
		}
		{
			GemmArgs g = plainGemm( c->dattn, m->at<f16>( e.wco ), M, d, d );
			g.epi = EPI_F32; g.bias = m->at<float>( e.bco ); g.res = c->dx; g.out32 = c->dx;
			WH_CHECK( product( g, nullptr, nullptr ) );
		}
		// MLP
		if( !fuseLn ) WH_CHECK( lnP( c, c->dx, m->at<float>( e.ln2w ), m->at<float>( e.ln2b ), c->dxn, M, d, true ) );
		{
			GemmArgs g = plainGemm( c->dxn, m->at<f16>( e.w1 ), M, 4 * d, d );
			g.epi = EPI_F16_GELU; g.bias = m->at<float>( e.b1 ); g.out16 = c->dh;
			WH_CHECK( product( g, m->at<float>( e.ln2w ), m->at<float>( e.ln2b ) ) );
		}
		{
			GemmArgs g = plainGemm( c->dh, m->at<f16>( e.w2 ), M, d, 4 * d );
			g.epi = EPI_F32; g.bias = m->at<float>( e.b2 ); g.res = c->dx; g.out32 = c->dx;
			WH_CHECK( product( g, nullptr, nullptr ) );
		}
	}
	// final norm + logits for the LAST token of every sequence only (the reference computes all rows, whisper.cpp:1840,
	// and then consumes just the last one, ContextImpl.cpp:159-169). The norm stays a separate launch here: fusing it
	// into the 3242 workgroups of the vocabulary product would re-read the rows 3242 times.
	WH_CHECK( lnP( c, c->dx, m->at<float>( L.decLnW ), m->at<float>( L.decLnB ), c->dxn, M, d, true ) );
	// prompts of different lengths (rows right-padded to nTokens): the row of every sequence's own last token moves to where the
	// product below reads, row nTokens - 1 of the sequence
	if( c->raggedLastPos && nTokens > 1 ) WH_CHECK( launchGatherLastRows( c->dxn, c->raggedLastPos, batch, nTokens, d, st ) );
	{
		GemmArgs g = plainGemm( c->dxn + (int64_t)( nTokens - 1 ) * d, m->at<f16>( L.te ), batch, hp.n_vocab, d );
		g.lda = nTokens * d;
		g.epi = EPI_F32; g.out32 = c->logits; g.ldc = hp.n_vocab;
		// more than 128 sequences: the vocabulary matrix (106 / 133 MB) against that many rows is a GEMM of ~50 GFLOP whose weight tiles should be
		// fetched once: the M-tiled kernel (128 x 128 tiles, 400 column tiles x ceil(batch / 128) row tiles)
		// (option vocab_decrows: gemmDecRows instead, for A/B runs)
		if( gemv && batch > GEMV_FUSED_MAX_ROWS && !g_opt.vocabDecRows )
		{
			g.nPastDev = nPastDev;
			WH_CHECK( gemmP( c, g, true, true ) );
		}
		else
			WH_CHECK( product( g, nullptr, nullptr ) );
	}
	return 0;
}

// Token ids index the embedding table: anything outside [0, n_vocab) is rejected at the boundary (the reference's addRows
// shader reads whatever the id selects, MlContext.cpp:588-618; here a bad id is the caller's error, not a device fault).
static int checkTokens( const wh_hparams& hp, const int32_t* tokens, int64_t count, const char* who )
{
	for( int64_t i = 0; i < count; i++ )
		if( tokens[ i ] < 0 || tokens[ i ] >= hp.n_vocab )
		{
			char buf[ 160 ];
			snprintf( buf, sizeof( buf ), "%s: token id %d at index %lld is outside [0, %d)", who, (int)tokens[ i ], (long long)i, hp.n_vocab );
			setError( buf );
			return WH_E_INVALIDARG;
		}
	return 0;
}

int wh_decode( wh_context* c, const int32_t* tokens, int batch, int nTokens, int nPast, float* logitsHost, float* probsHost )
{
	if( !c || !tokens || batch <= 0 || batch > c->maxSeq || ( batch % c->hyp ) != 0 || nTokens <= 0 || nPast < 0 ) { setError( "decode: bad argument" ); return WH_E_INVALIDARG; }
	if( !c->encoded ) { setError( "decode: wh_encode has not run" ); return WH_E_NOT_READY; }
	WH_BIND( c->m );
	const wh_hparams& hp = c->m->hp;
	if( nPast + nTokens > hp.n_text_ctx ) { setError( "decode: n_past + n_tokens exceeds n_text_ctx" ); return WH_E_BOUNDS; }
	hipStream_t st = c->stream;
	const int M = batch * nTokens;
	WH_CHECK( checkTokens( hp, tokens, M, "decode" ) );
	WH_HIP( hipMemcpyAsync( c->tokensDev, tokens, sizeof( int32_t ) * M, hipMemcpyHostToDevice, st ) );
	if( c->flags & WH_FLAG_PARITY_EXACT )
		WH_CHECK( decodeExact( c, batch, nTokens, nPast ) );
	else
	{
	WH_CHECK( decodeGraph( c, batch, nTokens, nPast, false ) );
	WH_CHECK( profiled( c, KC_SOFTMAX, 10.0 * batch * hp.n_vocab, 12.0 * batch * hp.n_vocab,
		[ & ]() { return launchVocabSoftMax( c->logits, c->probs, batch, hp.n_vocab, st ); } ) );
	}
	c->lastBatch = batch;
	if( logitsHost ) WH_HIP( hipMemcpyAsync( logitsHost, c->logits, sizeof( float ) * batch * hp.n_vocab, hipMemcpyDeviceToHost, st ) );
	if( probsHost ) WH_HIP( hipMemcpyAsync( probsHost, c->probs, sizeof( float ) * batch * hp.n_vocab, hipMemcpyDeviceToHost, st ) );
	if( logitsHost || probsHost ) WH_HIP( hipStreamSynchronize( st ) );
	return 0;
}

// The sampler state and the positions of `batch` sequences -> device, through the pinned staging (asynchronous; the staging is
// reused by the next call, which callers order behind a synchronisation of their own). positions == nullptr: `uniform` for all.
static int uploadDecodeState( wh_context* c, int batch, const DecodeState& s, const int32_t* positions, int uniform )
{
	DecodeState* const stState = (DecodeState*)( c->pinned + wh_context::PIN_STATE );
	int32_t* const stPos = c->pinned + wh_context::PIN_POS;
	static_assert( sizeof( DecodeState ) <= sizeof( int32_t ) * ( wh_context::PIN_POS - wh_context::PIN_STATE ), "staging layout" );
	*stState = s;
	for( int i = 0; i < batch; i++ ) stPos[ i ] = positions ? positions[ i ] : uniform;
	WH_HIP( hipMemcpyAsync( c->state, stState, sizeof( DecodeState ), hipMemcpyHostToDevice, c->stream ) );
	WH_HIP( hipMemcpyAsync( c->seqPos, stPos, sizeof( int32_t ) * batch, hipMemcpyHostToDevice, c->stream ) );
	return 0;
}

// logits -> table softmax -> sampleBest -> token data + next token: one workgroup per row, or -- for the few rows of one stream -- every row cut
// into 64 slices over the chip (three small launches, ~12 us instead of 41 on one CU)
static int sampleStep( wh_context* c, int batch, hipStream_t st )
{
	const wh_hparams& hp = c->m->hp;
	const SpecialIds sp = specialIds( hp );
	const int sot = sp.sot, solm = sp.solm, tnot = sp.tnot, beg = sp.beg;
	if( batch <= SMALL_MAX_ROWS && ( g_tuning & TUNE_SAMPLE_SPREAD ) )
	{
		if( !c->sampleScratch ) WH_CHECK( c->alloc( c->sampleScratch, (int64_t)sampleSpreadScratchBytes( SMALL_MAX_ROWS ), wh_context::DONT_CARE, "sampleScratch" ) );
		return profiled( c, KC_SAMPLE, 12.0 * batch * hp.n_vocab, 8.0 * batch * hp.n_vocab,
			[ & ]() { return launchSoftMaxSampleSpread( c->logits, c->probs, batch, hp.n_vocab, beg, sot, solm, tnot, c->state, c->greedyOut, c->tokensDev, c->mailDev, c->sampleScratch, st ); } );
	}
	return profiled( c, KC_SAMPLE, 12.0 * batch * hp.n_vocab, 8.0 * batch * hp.n_vocab,
		[ & ]() { return launchSoftMaxSample( c->logits, c->probs, batch, hp.n_vocab, beg, sot, solm, tnot, c->state, c->greedyOut, c->tokensDev, c->mailDev, st ); } );
}

// One greedy token on the device: decoder graph -> softmax + sampleBest -> advance the device-resident state.
static int greedyStep( wh_context* c, int batch )
{
	const wh_hparams& hp = c->m->hp;
	const SpecialIds sp = specialIds( hp );
	const int sot = sp.sot, solm = sp.solm, tnot = sp.tnot, beg = sp.beg;
	WH_CHECK( decodeGraph( c, batch, 1, 0, true ) );
	WH_CHECK( sampleStep( c, batch, c->stream ) );
	return launchAdvanceState( c->state, c->seqPos, batch, c->stream );
}

int wh_decode_greedy( wh_context* c, int batch, const int32_t* firstTokens, int nPast, int nSteps, int forceFirstTimestamp, int firstIsInitial,
	wh_token_data* out )
{
	if( !c || !firstTokens || !out || batch <= 0 || batch > c->maxSeq || ( batch % c->hyp ) != 0 || nSteps <= 0 || nPast < 0 ) { setError( "decode_greedy: bad argument" ); return WH_E_INVALIDARG; }
	if( !c->encoded ) { setError( "decode_greedy: wh_encode has not run" ); return WH_E_NOT_READY; }
	WH_BIND( c->m );
	const wh_hparams& hp = c->m->hp;
	if( nPast + nSteps > hp.n_text_ctx ) { setError( "decode_greedy: n_past + n_steps exceeds n_text_ctx" ); return WH_E_BOUNDS; }
	WH_CHECK( checkTokens( hp, firstTokens, batch, "decode_greedy" ) );
	hipStream_t st = c->stream;
	const DecodeState init = { 0, forceFirstTimestamp ? 1 : 0, firstIsInitial ? 1 : 0, 0 };
	WH_HIP( hipStreamSynchronize( st ) );	 // the staging may still be read by an earlier enqueue
	WH_CHECK( uploadDecodeState( c, batch, init, nullptr, nPast ) );
	WH_HIP( hipMemcpyAsync( c->tokensDev, firstTokens, sizeof( int32_t ) * batch, hipMemcpyHostToDevice, st ) );
	WH_HIP( hipStreamSynchronize( st ) );

	const bool useGraph = !c->prof.on && !( c->flags & WH_FLAG_NO_GRAPH ) && !debugSync();
	if( useGraph )
	{
		const uint32_t key = ( c->flags & WH_FLAG_PARITY_PV ) ? ( 0x10000u | (uint32_t)c->parityThreads ) : 0u;
		if( c->graphExec && ( c->graphBatch != batch || c->graphKey != key ) )
		{
			(void)hipGraphExecDestroy( c->graphExec );
			c->graphExec = nullptr;
		}
		if( !c->graphExec )
		{
			// one eager step first: it sets the per-kernel function attributes, which capture does not allow
			WH_CHECK( greedyStep( c, batch ) );
			WH_HIP( hipStreamSynchronize( st ) );
			WH_CHECK( uploadDecodeState( c, batch, init, nullptr, nPast ) );
			WH_HIP( hipMemcpyAsync( c->tokensDev, firstTokens, sizeof( int32_t ) * batch, hipMemcpyHostToDevice, st ) );
			WH_HIP( hipStreamSynchronize( st ) );
			hipGraph_t graph = nullptr;
			WH_HIP( hipStreamBeginCapture( st, hipStreamCaptureModeThreadLocal ) );
			const int rc = greedyStep( c, batch );
			const hipError_t e = hipStreamEndCapture( st, &graph );
			if( rc != 0 ) { if( graph ) (void)hipGraphDestroy( graph ); return rc; }
			if( e != hipSuccess ) return hipFail( e, "hipStreamEndCapture", __FILE__, __LINE__ );
			const hipError_t e2 = hipGraphInstantiate( &c->graphExec, graph, nullptr, nullptr, 0 );
			(void)hipGraphDestroy( graph );
			if( e2 != hipSuccess ) { c->graphExec = nullptr; return hipFail( e2, "hipGraphInstantiate", __FILE__, __LINE__ ); }
			c->graphBatch = batch;
			c->graphKey = key;
		}
		for( int s = 0; s < nSteps; s++ ) WH_HIP( hipGraphLaunch( c->graphExec, st ) );
	}
	else
	{
		for( int s = 0; s < nSteps; s++ )
		{
			c->profKeysHint = nPast + s + 1;
			WH_CHECK( greedyStep( c, batch ) );
		}
	}
	c->lastBatch = batch;
	static_assert( sizeof( wh_token_data ) == sizeof( TokenData ), "token data layout" );
	WH_HIP( hipMemcpyAsync( out, c->greedyOut, sizeof( TokenData ) * (size_t)batch * nSteps, hipMemcpyDeviceToHost, st ) );
	WH_HIP( hipStreamSynchronize( st ) );
	return 0;
}

// Records an event behind everything enqueued for the window so far
static int markWindow( wh_context* c )
{
	hipEvent_t e = nullptr;
	if( !c->markPool.empty() ) { e = c->markPool.back(); c->markPool.pop_back(); }
	else WH_HIP( hipEventCreateWithFlags( &e, hipEventDisableTiming ) );
	WH_HIP( hipEventRecord( e, c->stream ) );
	c->marks.push_back( { c->windowSamples, e } );
	return 0;
}

// Enqueues, without ever blocking the host: prompt step -> first sample (sampleTimestamp rules when requested) -> nSteps
// captured greedy steps. Token data of the 1 + nSteps samples stay on the device until wh_decode_window_finish.
// Several contexts driven this way from one host thread overlap on the GPU (each owns a stream): single-token decode
// steps are latency-bound and use a fraction of the chip, so independent windows fill it concurrently.
static int windowStart( wh_context* c, int batch, const int32_t* promptTokens, const int32_t* promptLens, int nPrompt, int nSteps, int forceFirstTimestamp,
	int firstIsInitial, const char* who )
{
	if( !c || !promptTokens || batch <= 0 || batch > c->maxSeq || ( batch % c->hyp ) != 0 || nPrompt <= 0 || nSteps < 0 ) { setError( "decode_window_start: bad argument" ); return WH_E_INVALIDARG; }
	if( !c->encoded ) { setError( "decode_window_start: wh_encode has not run" ); return WH_E_NOT_READY; }
	WH_BIND( c->m );
	const wh_hparams& hp = c->m->hp;
	if( nPrompt + nSteps > hp.n_text_ctx || (int64_t)batch * nPrompt > c->pinTokenCap() ) { setError( "decode_window_start: too many tokens" ); return WH_E_BOUNDS; }
	bool ragged = false;
	if( promptLens )
		for( int b = 0; b < batch; b++ )
		{
			if( promptLens[ b ] < 1 || promptLens[ b ] > nPrompt ) { setError( "decode_window_start: a prompt length is outside [1, nPromptMax]" ); return WH_E_INVALIDARG; }
			ragged = ragged || promptLens[ b ] != nPrompt;
		}
	if( ragged && c->hyp != 1 ) { setError( "decode_window_start: prompts of different lengths need one hypothesis per window" ); return WH_E_INVALIDARG; }
	// staging: the prompt tokens (padding of a shorter row = token 0: computed, never consumed), the sampler state, the positions
	int32_t* const stTok = c->pinTokens();
	const int M = batch * nPrompt;
	for( int b = 0; b < batch; b++ )
	{
		const int len = promptLens ? promptLens[ b ] : nPrompt;
		WH_CHECK( checkTokens( hp, promptTokens + (size_t)b * nPrompt, len, who ) );
		for( int i = 0; i < nPrompt; i++ ) stTok[ (size_t)b * nPrompt + i ] = i < len ? promptTokens[ (size_t)b * nPrompt + i ] : 0;
	}
	hipStream_t st = c->stream;
	for( auto& mk : c->marks ) c->markPool.push_back( mk.ev );
	c->marks.clear();
	const bool useGraph = !c->prof.on && !( c->flags & WH_FLAG_NO_GRAPH ) && nSteps > 0 && !debugSync();
	const uint32_t key = ( c->flags & WH_FLAG_PARITY_PV ) ? ( 0x10000u | (uint32_t)c->parityThreads ) : 0u;
	if( useGraph && c->graphExec && ( c->graphBatch != batch || c->graphKey != key ) )
	{
		WH_HIP( hipStreamSynchronize( st ) );
		(void)hipGraphExecDestroy( c->graphExec );
		c->graphExec = nullptr;
	}
	if( useGraph && !c->graphExec )
	{
		// first use for this batch size: one eager step (sets the per-kernel function attributes), then capture. Blocking,
		// once per context; the state it leaves behind is overwritten below.
		const DecodeState warm = { 0, 0, 0, 0 };
		WH_HIP( hipStreamSynchronize( st ) );
		{
			// not through the staging: it already holds this window's tokens
			std::vector<int32_t> zeros( (size_t)batch, 0 );
			WH_HIP( hipMemcpy( c->state, &warm, sizeof( warm ), hipMemcpyHostToDevice ) );
			WH_HIP( hipMemcpy( c->seqPos, zeros.data(), sizeof( int32_t ) * batch, hipMemcpyHostToDevice ) );
		}
		WH_HIP( hipMemsetAsync( c->tokensDev, 0, sizeof( int32_t ) * batch, st ) );
		WH_HIP( hipStreamSynchronize( st ) );
		WH_CHECK( greedyStep( c, batch ) );
		WH_HIP( hipStreamSynchronize( st ) );
		hipGraph_t graph = nullptr;
		WH_HIP( hipStreamBeginCapture( st, hipStreamCaptureModeThreadLocal ) );
		const int rc = greedyStep( c, batch );
		const hipError_t e = hipStreamEndCapture( st, &graph );
		if( rc != 0 ) { if( graph ) (void)hipGraphDestroy( graph ); return rc; }
		if( e != hipSuccess ) return hipFail( e, "hipStreamEndCapture", __FILE__, __LINE__ );
		const hipError_t e2 = hipGraphInstantiate( &c->graphExec, graph, nullptr, nullptr, 0 );
		(void)hipGraphDestroy( graph );
		if( e2 != hipSuccess ) { c->graphExec = nullptr; return hipFail( e2, "hipGraphInstantiate", __FILE__, __LINE__ ); }
		c->graphBatch = batch;
		c->graphKey = key;
	}
	// a new generation per window: stamps of earlier windows in the mailbox can never be taken for this one's
	// (only for the few sequences of a stream whose host loop reads every sample; a lock-step batch is read chunk by chunk through events)
	const bool mailbox = c->mailData && batch <= SMALL_MAX_ROWS;
	if( mailbox ) c->mailCounter = c->mailCounter == 0x7fffffff ? 1 : c->mailCounter + 1;
	c->mailGen = mailbox ? c->mailCounter : 0;
	// after the sampler's advance the position of sequence b must be its prompt length: start one below it
	const DecodeState s0 = { 0, forceFirstTimestamp ? 1 : 0, firstIsInitial ? 1 : 0, c->mailGen };
	{
		int32_t* const stPos = c->pinned + wh_context::PIN_POS;
		for( int b = 0; b < batch; b++ ) stPos[ b ] = ( promptLens ? promptLens[ b ] : nPrompt ) - 1;
		WH_CHECK( uploadDecodeState( c, batch, s0, stPos, 0 ) );
	}
	WH_HIP( hipMemcpyAsync( c->tokensDev, stTok, sizeof( int32_t ) * M, hipMemcpyHostToDevice, st ) );
	c->raggedLastPos = ragged ? c->seqPos : nullptr;
	const int rcGraph = decodeGraph( c, batch, nPrompt, 0, false );
	c->raggedLastPos = nullptr;
	WH_CHECK( rcGraph );
	{
		const SpecialIds sp = specialIds( hp );
		const int sot = sp.sot, solm = sp.solm, tnot = sp.tnot, beg = sp.beg;
		WH_CHECK( sampleStep( c, batch, st ) );
		WH_CHECK( launchAdvanceState( c->state, c->seqPos, batch, st ) );
	}
	if( useGraph )
		for( int s = 0; s < nSteps; s++ ) WH_HIP( hipGraphLaunch( c->graphExec, st ) );
	else
		for( int s = 0; s < nSteps; s++ )
		{
			c->profKeysHint = nPrompt + s + 1;
			WH_CHECK( greedyStep( c, batch ) );
		}
	c->lastBatch = batch;
	c->windowSamples = 1 + nSteps;
	c->windowPos = nPrompt + nSteps;
	return markWindow( c );
}

int wh_decode_window_start( wh_context* c, int batch, const int32_t* promptTokens, int nPrompt, int nSteps, int forceFirstTimestamp, int firstIsInitial )
{
	return windowStart( c, batch, promptTokens, nullptr, nPrompt, nSteps, forceFirstTimestamp, firstIsInitial, "decode_window_start" );
}

int wh_decode_window_start_ragged( wh_context* c, int batch, const int32_t* promptTokens, const int32_t* promptLens, int nPromptMax, int nSteps,
	int forceFirstTimestamp, int firstIsInitial )
{
	if( !promptLens ) { setError( "decode_window_start_ragged: promptLens is null" ); return WH_E_INVALIDARG; }
	return windowStart( c, batch, promptTokens, promptLens, nPromptMax, nSteps, forceFirstTimestamp, firstIsInitial, "decode_window_start_ragged" );
}

// More greedy steps of the window wh_decode_window_start began, still without blocking the host: the position, the last
// token and the sampler flags live in device memory, so nothing has to come back first. A host loop that looks for a
// stop token keeps one chunk queued behind the one it is reading (wh_decode_window_fetch) and the GPU never idles.
int wh_decode_window_continue( wh_context* c, int nSteps )
{
	if( !c || nSteps <= 0 || c->windowSamples <= 0 ) { setError( "decode_window_continue: no window in progress" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	const wh_hparams& hp = c->m->hp;
	if( c->windowPos + nSteps > hp.n_text_ctx ) { setError( "decode_window_continue: n_text_ctx exceeded" ); return WH_E_BOUNDS; }
	const int batch = c->lastBatch;
	const bool useGraph = !c->prof.on && !( c->flags & WH_FLAG_NO_GRAPH ) && !debugSync();
	const uint32_t key = ( c->flags & WH_FLAG_PARITY_PV ) ? ( 0x10000u | (uint32_t)c->parityThreads ) : 0u;
	if( useGraph && ( !c->graphExec || c->graphBatch != batch || c->graphKey != key ) )
	{
		// the window was started with nSteps == 0 (no graph yet): capture now. The eager warm-up step must not disturb
		// the window's device state, so the state and the pending token are saved around it.
		WH_HIP( hipStreamSynchronize( c->stream ) );
		DecodeState saved;
		std::vector<int32_t> tok( (size_t)batch ), savedPos( (size_t)batch );
		WH_HIP( hipMemcpy( &saved, c->state, sizeof( saved ), hipMemcpyDeviceToHost ) );
		WH_HIP( hipMemcpy( savedPos.data(), c->seqPos, sizeof( int32_t ) * batch, hipMemcpyDeviceToHost ) );
		WH_HIP( hipMemcpy( tok.data(), c->tokensDev, sizeof( int32_t ) * batch, hipMemcpyDeviceToHost ) );
		if( c->graphExec ) { (void)hipGraphExecDestroy( c->graphExec ); c->graphExec = nullptr; }
		WH_CHECK( greedyStep( c, batch ) );
		WH_HIP( hipStreamSynchronize( c->stream ) );
		hipGraph_t graph = nullptr;
		WH_HIP( hipStreamBeginCapture( c->stream, hipStreamCaptureModeThreadLocal ) );
		const int rc = greedyStep( c, batch );
		const hipError_t e = hipStreamEndCapture( c->stream, &graph );
		if( rc != 0 ) { if( graph ) (void)hipGraphDestroy( graph ); return rc; }
		if( e != hipSuccess ) return hipFail( e, "hipStreamEndCapture", __FILE__, __LINE__ );
		const hipError_t e2 = hipGraphInstantiate( &c->graphExec, graph, nullptr, nullptr, 0 );
		(void)hipGraphDestroy( graph );
		if( e2 != hipSuccess ) { c->graphExec = nullptr; return hipFail( e2, "hipGraphInstantiate", __FILE__, __LINE__ ); }
		c->graphBatch = batch;
		c->graphKey = key;
		WH_HIP( hipMemcpy( c->state, &saved, sizeof( saved ), hipMemcpyHostToDevice ) );
		WH_HIP( hipMemcpy( c->seqPos, savedPos.data(), sizeof( int32_t ) * batch, hipMemcpyHostToDevice ) );
		WH_HIP( hipMemcpy( c->tokensDev, tok.data(), sizeof( int32_t ) * batch, hipMemcpyHostToDevice ) );
	}
	if( useGraph )
		for( int s = 0; s < nSteps; s++ ) WH_HIP( hipGraphLaunch( c->graphExec, c->stream ) );
	else
		for( int s = 0; s < nSteps; s++ )
		{
			c->profKeysHint = c->windowPos + s + 1;
			WH_CHECK( greedyStep( c, batch ) );
		}
	c->windowSamples += nSteps;
	c->windowPos += nSteps;
	return markWindow( c );
}

// Samples [first, first + count) of the window in progress, HOST [count][batch]; blocks only until THOSE samples exist
// (chunks enqueued behind them keep running).
int wh_decode_window_fetch( wh_context* c, int first, int count, wh_token_data* out )
{
	if( !c || !out || first < 0 || count <= 0 || first + count > c->windowSamples ) { setError( "decode_window_fetch: range not enqueued" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	const wh_context::Mark* mk = nullptr;
	for( const auto& x : c->marks )
		if( x.endSample >= first + count ) { mk = &x; break; }
	if( !mk ) { setError( "decode_window_fetch: no completion mark for that range" ); return WH_E_INVALIDARG; }
	if( c->mailData && c->mailGen != 0 && !c->prof.on )
	{
		// the sampler stamps the pinned mailbox after each sample: poll the last stamp of the range (samples are produced in
		// order on one stream, rows of a step by concurrent workgroups: every row of the last step is checked). Nothing is
		// enqueued anywhere -- an event wait + copy + synchronise costs the DECODE stream ~0.2 ms per step (measured, decode1_prof).
		const size_t rows = (size_t)c->lastBatch;
		const volatile int* const flag = c->mailFlag;
		const TokenData* const data = c->mailData + (size_t)first * rows;
		const size_t slot0 = (size_t)first * rows, nRec = rows * (size_t)count;
		const auto t0 = std::chrono::steady_clock::now();
		auto checksum = [ & ]( const TokenData& r ) -> int
		{
			int p, pt, ps;
			memcpy( &p, &r.p, 4 ); memcpy( &pt, &r.pt, 4 ); memcpy( &ps, &r.ptsum, 4 );
			return (int)( (unsigned)c->mailGen ^ (unsigned)r.id ^ ( (unsigned)r.tid * 0x9E3779B1u ) ^ (unsigned)p ^ ( (unsigned)pt * 3u ) ^ ( (unsigned)ps * 5u ) );
		};
		bool ready = false;
		for( long spins = 0; !ready; spins++ )
		{
			ready = true;
			for( size_t r = 0; r < nRec && ready; r++ ) ready = flag[ 2 * ( slot0 + r ) ] == c->mailGen;
			if( ready )
			{
				// stamped: take the records and hold each against its checksum (a torn or early read fails it: poll on)
				std::atomic_thread_fence( std::memory_order_acquire );
				memcpy( out, data, sizeof( TokenData ) * nRec );
				for( size_t r = 0; r < nRec && ready; r++ )
				{
					TokenData td;
					memcpy( &td, (const char*)out + r * sizeof( TokenData ), sizeof( td ) );
					ready = flag[ 2 * ( slot0 + r ) + 1 ] == checksum( td );
				}
				if( ready ) return 0;
			}
			if( ( spins & 255 ) == 255 )
			{
				// a stamp that does not arrive (a driver that does not make the mailbox visible) must not hang the caller;
				// a long wait (the encoder of a window runs ahead of its first sample) does not need a spinning core
				const double waited = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
				if( waited > 0.25 ) break;
				if( waited > 0.002 ) std::this_thread::sleep_for( std::chrono::microseconds( 20 ) );
				else std::this_thread::yield();
			}
		}
	}
	if( !c->copyStream ) WH_HIP( hipStreamCreateWithFlags( &c->copyStream, hipStreamNonBlocking ) );
	WH_HIP( hipStreamWaitEvent( c->copyStream, mk->ev, 0 ) );
	WH_HIP( hipMemcpyAsync( out, c->greedyOut + (size_t)first * c->lastBatch, sizeof( TokenData ) * (size_t)count * c->lastBatch, hipMemcpyDeviceToHost, c->copyStream ) );
	WH_HIP( hipStreamSynchronize( c->copyStream ) );
	return 0;
}

int wh_decode_window_ready( wh_context* c, int first, int count )
{
	if( !c || first < 0 || count <= 0 || first + count > c->windowSamples ) { setError( "decode_window_ready: range not enqueued" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	for( const auto& x : c->marks )
		if( x.endSample >= first + count )
		{
			const hipError_t e = hipEventQuery( x.ev );
			if( e == hipSuccess ) return 1;
			if( e == hipErrorNotReady ) { (void)hipGetLastError(); return 0; }
			return hipFail( e, "hipEventQuery", __FILE__, __LINE__ );
		}
	setError( "decode_window_ready: no completion mark for that range" );
	return WH_E_INVALIDARG;
}

int wh_decode_window_finish( wh_context* c, wh_token_data* out )
{
	if( !c || !out || c->windowSamples <= 0 ) { setError( "decode_window_finish: nothing was started" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	WH_HIP( hipMemcpyAsync( out, c->greedyOut, sizeof( TokenData ) * (size_t)c->lastBatch * c->windowSamples, hipMemcpyDeviceToHost, c->stream ) );
	WH_HIP( hipStreamSynchronize( c->stream ) );
	c->windowSamples = 0;
	return 0;
}

int wh_sample_best( wh_context* c, int batch, int forceTimestamp, int isInitial, wh_token_data* out )
{
	if( !c || !out || batch <= 0 || batch > c->maxSeq ) { setError( "sample_best: bad argument" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	const wh_hparams& hp = c->m->hp;
	const SpecialIds sp = specialIds( hp );
	const int sot = sp.sot, solm = sp.solm, tnot = sp.tnot, beg = sp.beg;
	WH_CHECK( profiled( c, KC_SAMPLE, 0.0, 5.0 * 4.0 * batch * hp.n_vocab,
		[ & ]() { return launchSampleBest( c->probs, batch, hp.n_vocab, beg, sot, solm, tnot, forceTimestamp, isInitial, c->tokDataDev, c->stream ); } ) );
	static_assert( sizeof( wh_token_data ) == sizeof( TokenData ), "token data layout" );
	WH_HIP( hipMemcpyAsync( out, c->tokDataDev, sizeof( TokenData ) * batch, hipMemcpyDeviceToHost, c->stream ) );
	WH_HIP( hipStreamSynchronize( c->stream ) );
	return 0;
}

int wh_beam_candidates( wh_context* c, int batch, int width, int forceTimestamp, int isInitial, wh_token_data* out )
{
	if( !c || !out || batch <= 0 || batch > c->maxSeq || width < 1 || width > 8 ) { setError( "beam_candidates: bad argument" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	const wh_hparams& hp = c->m->hp;
	const SpecialIds sp = specialIds( hp );
	if( !c->beamCand ) WH_CHECK( c->alloc( c->beamCand, (int64_t)c->maxSeq * 8, wh_context::DONT_CARE, "beamCand" ) );
	WH_CHECK( profiled( c, KC_SAMPLE, 0.0, ( 4.0 + width ) * 4.0 * batch * hp.n_vocab,
		[ & ]() { return launchBeamCandidates( c->probs, batch, hp.n_vocab, sp.beg, sp.sot, sp.solm, sp.tnot, forceTimestamp, isInitial, width, c->beamCand, c->stream ); } ) );
	WH_HIP( hipMemcpyAsync( out, c->beamCand, sizeof( TokenData ) * (size_t)batch * width, hipMemcpyDeviceToHost, c->stream ) );
	WH_HIP( hipStreamSynchronize( c->stream ) );
	return 0;
}

int wh_reorder_self_cache( wh_context* c, int batch, const int32_t* parents, int rows )
{
	if( !c || !parents || batch <= 0 || batch > c->maxSeq || rows < 0 ) { setError( "reorder_self_cache: bad argument" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	const wh_hparams& hp = c->m->hp;
	if( rows > hp.n_text_ctx ) { setError( "reorder_self_cache: more rows than n_text_ctx" ); return WH_E_BOUNDS; }
	bool any = false;
	for( int j = 0; j < batch; j++ )
	{
		if( parents[ j ] < 0 || parents[ j ] >= batch ) { setError( "reorder_self_cache: a parent is outside the batch" ); return WH_E_INVALIDARG; }
		any = any || parents[ j ] != j;
	}
	if( !any || rows == 0 ) return 0;
	const int64_t n = (int64_t)hp.n_text_layer * c->maxSeq * hp.n_text_ctx * hp.n_text_state;
	if( !c->selfKScratch ) WH_CHECK( c->alloc( c->selfKScratch, n, wh_context::DONT_CARE, "selfKScratch" ) );
	if( !c->selfVScratch ) WH_CHECK( c->alloc( c->selfVScratch, n, wh_context::DONT_CARE, "selfVScratch" ) );
	WH_HIP( hipStreamSynchronize( c->stream ) );	   // the staging below may still be read by an earlier enqueue
	int32_t* const st = c->pinned + wh_context::PIN_POS;
	for( int j = 0; j < batch; j++ ) st[ j ] = parents[ j ];
	WH_HIP( hipMemcpyAsync( c->tokDataDev, st, sizeof( int32_t ) * batch, hipMemcpyHostToDevice, c->stream ) );	  // tokDataDev: [maxSeq] records of 20 bytes, free between samples
	return profiled( c, KC_EMBED, 0.0, 4.0 * 2.0 * 2.0 * rows * hp.n_text_state * hp.n_text_layer * batch,
		[ & ]() { return launchReorderCache( c->selfK, c->selfV, c->selfKScratch, c->selfVScratch, (const int*)c->tokDataDev, hp.n_text_layer, batch, c->maxSeq,
			hp.n_text_head, hp.n_text_ctx, rows, c->stream ); } );
}

// ---- beam search with the ranking on the device: no host round trip between the steps of a window ----
static int beamBuffers( wh_context* c )
{
	const wh_hparams& hp = c->m->hp;
	if( !c->beamCand ) WH_CHECK( c->alloc( c->beamCand, (int64_t)c->maxSeq * 8, wh_context::DONT_CARE, "beamCand" ) );
	const int64_t n = (int64_t)hp.n_text_layer * c->maxSeq * hp.n_text_ctx * hp.n_text_state;
	if( !c->selfKScratch ) WH_CHECK( c->alloc( c->selfKScratch, n, wh_context::DONT_CARE, "selfKScratch" ) );
	if( !c->selfVScratch ) WH_CHECK( c->alloc( c->selfVScratch, n, wh_context::DONT_CARE, "selfVScratch" ) );
	if( !c->beamRules ) WH_CHECK( c->alloc( c->beamRules, c->maxBatch, wh_context::MUST_BE_ZERO, "beamRules" ) );
	if( !c->beamState ) WH_CHECK( c->alloc( c->beamState, c->maxBatch, wh_context::MUST_BE_ZERO, "beamState" ) );
	if( !c->beamRecords ) WH_CHECK( c->alloc( c->beamRecords, (int64_t)hp.n_text_ctx * c->maxBatch * BEAM_MAX_WIDTH, wh_context::DONT_CARE, "beamRecords" ) );
	if( !c->beamParents ) WH_CHECK( c->alloc( c->beamParents, c->maxSeq, wh_context::MUST_BE_ZERO, "beamParents" ) );
	return 0;
}

// one step: parents' cache rows move -> every slot decodes its token -> probabilities -> candidates -> ranking (writes the next step's parents and tokens)
static int beamStep( wh_context* c, int batch, int width )
{
	const wh_hparams& hp = c->m->hp;
	const SpecialIds sp = specialIds( hp );
	hipStream_t st = c->stream;
	WH_CHECK( profiled( c, KC_EMBED, 0.0, 4.0 * 2.0 * 2.0 * c->profKeysHint * hp.n_text_state * hp.n_text_layer * batch,
		[ & ]() { return launchReorderCacheDev( c->selfK, c->selfV, c->selfKScratch, c->selfVScratch, c->beamParents, c->seqPos, hp.n_text_layer, batch, c->maxSeq,
			hp.n_text_head, hp.n_text_ctx, c->hyp, st ); } ) );
	WH_CHECK( decodeGraph( c, batch, 1, 0, true ) );
	WH_CHECK( profiled( c, KC_SOFTMAX, 10.0 * batch * hp.n_vocab, 12.0 * batch * hp.n_vocab, [ & ]() { return launchVocabSoftMax( c->logits, c->probs, batch, hp.n_vocab, st ); } ) );
	WH_CHECK( profiled( c, KC_SAMPLE, 0.0, ( 4.0 + width ) * 4.0 * batch * hp.n_vocab,
		[ & ]() { return launchBeamCandidates( c->probs, batch, hp.n_vocab, sp.beg, sp.sot, sp.solm, sp.tnot, 0, 0, width, c->beamCand, st ); } ) );
	WH_CHECK( profiled( c, KC_SAMPLE, 0.0, 0.0, [ & ]() { return launchBeamRank( c->beamCand, batch / c->hyp, c->hyp, width, c->beamRules, c->beamState, c->beamRecords,
		hp.n_text_ctx, c->beamParents, c->tokensDev, st ); } ) );
	return launchAdvanceState( c->state, c->seqPos, batch, st );
}

static int beamEnqueue( wh_context* c, int nSteps )
{
	const int batch = c->beamWindows * c->hyp, width = c->beamWidth;
	// the graph wh_beam_window_start captured for this shape; eager launches when there is none (profiler on, WH_FLAG_NO_GRAPH, WH_DEBUG_SYNC)
	const bool useGraph = !c->prof.on && !( c->flags & WH_FLAG_NO_GRAPH ) && !debugSync() && c->beamGraphExec && c->beamGraphBatch == batch && c->beamGraphWidth == width;
	for( int s = 0; s < nSteps; s++ )
	{
		if( useGraph ) WH_HIP( hipGraphLaunch( c->beamGraphExec, c->stream ) );
		else
		{
			c->profKeysHint = c->windowPos + s + 1;
			WH_CHECK( beamStep( c, batch, width ) );
		}
	}
	c->beamSteps += nSteps;
	c->windowPos += nSteps;
	return 0;
}

int wh_beam_window_start( wh_context* c, int windows, const int32_t* promptTokens, int nPrompt, int width, const wh_beam_rules* rules, int nSteps )
{
	if( !c || !promptTokens || !rules || windows <= 0 || windows > c->maxBatch || nPrompt <= 0 || nSteps < 0 || width < 1 || width > c->hyp || width > BEAM_MAX_WIDTH )
	{
		setError( "beam_window_start: bad argument (1 <= width <= hypotheses per window of the context <= 8)" );
		return WH_E_INVALIDARG;
	}
	if( !c->encoded ) { setError( "beam_window_start: wh_encode has not run" ); return WH_E_NOT_READY; }
	WH_BIND( c->m );
	const wh_hparams& hp = c->m->hp;
	const int batch = windows * c->hyp;
	if( nPrompt + nSteps > hp.n_text_ctx || (int64_t)batch * nPrompt > c->pinTokenCap() ) { setError( "beam_window_start: too many tokens" ); return WH_E_BOUNDS; }
	static_assert( sizeof( wh_beam_rules ) == sizeof( BeamRules ) && sizeof( wh_beam_window ) == sizeof( BeamWindow ) && sizeof( wh_beam_record ) == sizeof( BeamRecord ) &&
		sizeof( wh_beam_hyp ) == sizeof( BeamHyp ), "beam structure layouts" );
	WH_CHECK( beamBuffers( c ) );
	hipStream_t st = c->stream;
	WH_HIP( hipStreamSynchronize( st ) );	   // the staging and the search state may still be read by an earlier window
	const SpecialIds sp = specialIds( hp );
	c->beamWindows = windows;
	c->beamWidth = width;
	c->beamSteps = 0;
	const bool useGraph = !c->prof.on && !( c->flags & WH_FLAG_NO_GRAPH ) && !debugSync();
	if( useGraph && ( !c->beamGraphExec || c->beamGraphBatch != batch || c->beamGraphWidth != width ) )
	{
		// first use for this shape: one eager step with a finished search (the ranking returns at once, parents = identity) sets the per-kernel function
		// attributes, then the capture. Blocking, once per context and shape; what it leaves behind is overwritten below.
		std::vector<BeamWindow> idle( (size_t)windows );
		memset( idle.data(), 0, idle.size() * sizeof( BeamWindow ) );
		for( BeamWindow& w : idle ) w.done = 1;
		std::vector<int32_t> zeros( (size_t)batch, 0 ), ident( (size_t)batch );
		for( int b = 0; b < batch; b++ ) ident[ (size_t)b ] = b;
		const DecodeState warm = { 0, 0, 0, 0 };
		WH_HIP( hipMemcpy( c->beamState, idle.data(), idle.size() * sizeof( BeamWindow ), hipMemcpyHostToDevice ) );
		WH_HIP( hipMemcpy( c->beamRules, rules, sizeof( BeamRules ) * windows, hipMemcpyHostToDevice ) );
		WH_HIP( hipMemcpy( c->beamParents, ident.data(), sizeof( int32_t ) * batch, hipMemcpyHostToDevice ) );
		WH_HIP( hipMemcpy( c->state, &warm, sizeof( warm ), hipMemcpyHostToDevice ) );
		WH_HIP( hipMemcpy( c->seqPos, zeros.data(), sizeof( int32_t ) * batch, hipMemcpyHostToDevice ) );
		WH_HIP( hipMemcpy( c->tokensDev, zeros.data(), sizeof( int32_t ) * batch, hipMemcpyHostToDevice ) );
		WH_CHECK( beamStep( c, batch, width ) );
		WH_HIP( hipStreamSynchronize( st ) );
		if( c->beamGraphExec ) { (void)hipGraphExecDestroy( c->beamGraphExec ); c->beamGraphExec = nullptr; }
		// the launch sequence of a step is the same for every token: positions, parents, tokens and the search state live in device memory
		{
			hipGraph_t graph = nullptr;
			WH_HIP( hipStreamBeginCapture( st, hipStreamCaptureModeThreadLocal ) );
			const int rc = beamStep( c, batch, width );
			const hipError_t e = hipStreamEndCapture( st, &graph );
			if( rc != 0 ) { if( graph ) (void)hipGraphDestroy( graph ); return rc; }
			if( e != hipSuccess ) return hipFail( e, "hipStreamEndCapture", __FILE__, __LINE__ );
			const hipError_t e2 = hipGraphInstantiate( &c->beamGraphExec, graph, nullptr, nullptr, 0 );
			(void)hipGraphDestroy( graph );
			if( e2 != hipSuccess ) { c->beamGraphExec = nullptr; return hipFail( e2, "hipGraphInstantiate", __FILE__, __LINE__ ); }
			c->beamGraphBatch = batch;
			c->beamGraphWidth = width;
		}
	}
	// the window's own state: rules, an empty search, the prompt of every slot, positions
	{
		std::vector<BeamWindow> init( (size_t)windows );
		memset( init.data(), 0, init.size() * sizeof( BeamWindow ) );
		for( BeamWindow& w : init ) { w.nPrompt = nPrompt; w.nTextCtx = hp.n_text_ctx; }
		WH_HIP( hipMemcpy( c->beamState, init.data(), init.size() * sizeof( BeamWindow ), hipMemcpyHostToDevice ) );
		WH_HIP( hipMemcpy( c->beamRules, rules, sizeof( BeamRules ) * windows, hipMemcpyHostToDevice ) );
	}
	int32_t* const stTok = c->pinTokens();
	const int M = batch * nPrompt;
	for( int w = 0; w < windows; w++ )
	{
		WH_CHECK( checkTokens( hp, promptTokens + (size_t)w * nPrompt, nPrompt, "beam_window_start" ) );
		for( int j = 0; j < c->hyp; j++ )
			for( int i = 0; i < nPrompt; i++ ) stTok[ ( (size_t)w * c->hyp + j ) * nPrompt + i ] = promptTokens[ (size_t)w * nPrompt + i ];
	}
	const DecodeState s0 = { 0, 0, 0, 0 };
	WH_CHECK( uploadDecodeState( c, batch, s0, nullptr, nPrompt ) );	   // after the prompt step every slot stands at position nPrompt
	WH_HIP( hipMemcpyAsync( c->tokensDev, stTok, sizeof( int32_t ) * M, hipMemcpyHostToDevice, st ) );
	WH_CHECK( decodeGraph( c, batch, nPrompt, 0, false ) );
	WH_CHECK( profiled( c, KC_SOFTMAX, 10.0 * batch * hp.n_vocab, 12.0 * batch * hp.n_vocab, [ & ]() { return launchVocabSoftMax( c->logits, c->probs, batch, hp.n_vocab, st ); } ) );
	// the first sample of a window: the reference's sampleTimestamp( true ) rules (forced timestamp, the 1.00 s cap)
	WH_CHECK( profiled( c, KC_SAMPLE, 0.0, ( 4.0 + width ) * 4.0 * batch * hp.n_vocab,
		[ & ]() { return launchBeamCandidates( c->probs, batch, hp.n_vocab, sp.beg, sp.sot, sp.solm, sp.tnot, 1, 1, width, c->beamCand, st ); } ) );
	WH_CHECK( profiled( c, KC_SAMPLE, 0.0, 0.0, [ & ]() { return launchBeamRank( c->beamCand, windows, c->hyp, width, c->beamRules, c->beamState, c->beamRecords, hp.n_text_ctx,
		c->beamParents, c->tokensDev, st ); } ) );
	c->beamSteps = 1;
	c->windowPos = nPrompt;
	c->lastBatch = batch;
	return beamEnqueue( c, nSteps );
}

int wh_beam_window_continue( wh_context* c, int nSteps )
{
	if( !c || nSteps <= 0 || c->beamSteps <= 0 ) { setError( "beam_window_continue: no window in progress" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	if( c->windowPos + nSteps > c->m->hp.n_text_ctx ) { setError( "beam_window_continue: n_text_ctx exceeded" ); return WH_E_BOUNDS; }
	return beamEnqueue( c, nSteps );
}

int wh_beam_window_status( wh_context* c, wh_beam_window* out )
{
	if( !c || !out || c->beamSteps <= 0 ) { setError( "beam_window_status: no window in progress" ); return WH_E_INVALIDARG; }
	WH_BIND( c->m );
	WH_HIP( hipMemcpyAsync( out, c->beamState, sizeof( BeamWindow ) * (size_t)c->beamWindows, hipMemcpyDeviceToHost, c->stream ) );
	WH_HIP( hipStreamSynchronize( c->stream ) );
	return 0;
}

int wh_beam_window_records( wh_context* c, int firstStep, int count, wh_beam_record* out )
{
	if( !c || !out || firstStep < 0 || count <= 0 || firstStep + count > c->beamSteps ) { setError( "beam_window_records: steps outside what was enqueued" ); return WH_E_BOUNDS; }
	WH_BIND( c->m );
	// device layout [step][windows of the call][width]
	const size_t perStep = (size_t)c->beamWindows * c->beamWidth;
	WH_HIP( hipMemcpyAsync( out, c->beamRecords + (size_t)firstStep * perStep, sizeof( BeamRecord ) * perStep * count, hipMemcpyDeviceToHost, c->stream ) );
	WH_HIP( hipStreamSynchronize( c->stream ) );
	return 0;
}

int wh_profile_enable( wh_context* c, int on )
{
	if( !c ) return WH_E_INVALIDARG;
	WH_BIND( c->m );
	c->prof.reset();
	c->prof.on = on != 0;
	if( c->prof.on )
	{
		// calibration: what an event pair around a launch measures when the kernel does nothing (the first launch of a
		// kernel loads its code object: keep that out of the average)
		for( int i = 0; i < 4; i++ ) hipLaunchKernelGGL( probeEmpty, dim3( 1 ), dim3( 64 ), 0, c->stream, (int*)nullptr );
		WH_HIP( hipStreamSynchronize( c->stream ) );
		for( int i = 0; i < 64; i++ )
			WH_CHECK( profiled( c, KC_EVENT_PAIR, 0.0, 0.0, [ & ]() { hipLaunchKernelGGL( probeEmpty, dim3( 1 ), dim3( 64 ), 0, c->stream, (int*)nullptr ); return 0; } ) );
	}
	return 0;
}

int wh_profile_read( wh_context* c, wh_profile_entry* out, int cap, int* count )
{
	if( !c || !count ) return WH_E_INVALIDARG;
	WH_BIND( c->m );
	WH_HIP( hipStreamSynchronize( c->stream ) );
	c->prof.resolve();
	int n = 0;
	for( int i = 0; i < KC_COUNT; i++ )
	{
		if( c->prof.calls[ i ] == 0 ) continue;
		if( out && n < cap )
		{
			wh_profile_entry& e = out[ n ];
			memset( &e, 0, sizeof( e ) );
			snprintf( e.name, sizeof( e.name ), "%s", kernelClassNames[ i ] );
			e.calls = c->prof.calls[ i ];
			e.ms = c->prof.ms[ i ];
			e.flops = c->prof.flops[ i ];
			e.bytes = c->prof.bytes[ i ];
		}
		n++;
	}
	*count = n;
	return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// debug reads
// ------------------------------------------------------------------------------------------------------------------
static int readHeadMajor( wh_context* c, const f16* src, int batch, int rows, int rowStride, float* dst )
{
	// [b][h][rowStride][64] FP16 -> [b][rows][H*64] FP32
	const int H = c->m->hp.n_audio_head, d = H * HEAD_DIM;
	std::vector<uint16_t> tmp( (size_t)batch * H * rowStride * HEAD_DIM );
	WH_HIP( hipStreamSynchronize( c->stream ) );
	WH_HIP( hipMemcpy( tmp.data(), src, tmp.size() * 2, hipMemcpyDeviceToHost ) );
	for( int b = 0; b < batch; b++ )
		for( int h = 0; h < H; h++ )
			for( int t = 0; t < rows; t++ )
				for( int j = 0; j < HEAD_DIM; j++ )
					dst[ ( (size_t)b * rows + t ) * d + h * HEAD_DIM + j ] = f16BitsToF32( tmp[ ( ( (size_t)b * H + h ) * rowStride + t ) * HEAD_DIM + j ] );
	return 0;
}

int wh_debug_read( wh_context* c, const char* what, int layer, int rows, float* dstHost, int64_t dstCapFloats )
{
	if( !c || !what || !dstHost ) return WH_E_INVALIDARG;
	WH_BIND( c->m );
	const wh_hparams& hp = c->m->hp;
	const int d = hp.n_audio_state;
	const int batch = c->lastEncBatch;
	const int seqs = c->lastBatch;
	const std::string w = what;
	if( w == "exp-table" )
	{
		// the arena's copy of the reference's exponential table (non-positive arguments), as FP32 values
		if( dstCapFloats < EXP_TABLE_ENTRIES ) return WH_E_BOUNDS;
		std::vector<uint16_t> tmp( EXP_TABLE_ENTRIES );
		WH_HIP( hipMemcpy( tmp.data(), c->m->at<f16>( c->m->L.expTab ), EXP_TABLE_ENTRIES * 2, hipMemcpyDeviceToHost ) );
		for( int i = 0; i < EXP_TABLE_ENTRIES; i++ ) dstHost[ i ] = f16BitsToF32( tmp[ (size_t)i ] );
		return 0;
	}
	if( w == "exact-gelu-table" || w == "exact-exp-table" )
	{
		// WH_FLAG_PARITY_EXACT's copies of ggml_init's tables, all 65536 entries, as FP32 values
		if( dstCapFloats < 65536 ) return WH_E_BOUNDS;
		WH_CHECK( exactTables( c ) );
		std::vector<uint16_t> tmp( 65536 );
		WH_HIP( hipMemcpy( tmp.data(), w == "exact-gelu-table" ? c->ex.gelu : c->ex.expt, 65536 * 2, hipMemcpyDeviceToHost ) );
		for( int i = 0; i < 65536; i++ ) dstHost[ i ] = f16BitsToF32( tmp[ (size_t)i ] );
		return 0;
	}
	if( w.compare( 0, 6, "exact:" ) == 0 )
	{
		// a buffer of the exact-order encoder as the last wh_encode left it (first chunk of windows): x, cur, q, k, v, kqv [windows][n_ctx][d], h [..][4 d], conv1 [..][2 n_ctx][d]
		const std::string n = w.substr( 6 );
		const wh_context::Exact& e = c->ex;
		const float* src = n == "x" ? e.x : n == "cur" ? e.cur : n == "q" ? e.q : n == "k" ? e.k : n == "v" ? e.v : n == "kqv" ? e.kqv : n == "h" ? e.h : n == "conv1" ? e.conv1 : nullptr;
		if( !src ) { setError( "debug_read: no such exact-mode buffer (or the mode has not run)" ); return WH_E_NOT_READY; }
		const int64_t count = (int64_t)std::min( batch, e.encWindows ) * c->T * d * ( n == "h" ? 4 : n == "conv1" ? 2 : 1 );
		if( dstCapFloats < count ) return WH_E_BOUNDS;
		WH_HIP( hipStreamSynchronize( c->stream ) );
		WH_HIP( hipMemcpy( dstHost, src, (size_t)count * 4, hipMemcpyDeviceToHost ) );
		return 0;
	}
	if( w == "logits" || w == "probs" )
	{
		// what the LAST decode step left: logits (or, after wh_decode with probsHost / the sampler, probabilities) of every sequence, [seqs][n_vocab]
		const int64_t n = (int64_t)seqs * hp.n_vocab;
		if( n <= 0 || dstCapFloats < n ) return WH_E_BOUNDS;
		WH_HIP( hipStreamSynchronize( c->stream ) );
		WH_HIP( hipMemcpy( dstHost, w == "logits" ? c->logits : c->probs, (size_t)n * 4, hipMemcpyDeviceToHost ) );
		return 0;
	}
	if( w == "cross-k1" || w == "cross-v1" )
	{
		// ONE window (index `rows`) of a layer's cross-attention cache: [n_ctx][d] -- contexts of hundreds of windows
		if( layer < 0 || layer >= hp.n_text_layer || rows < 0 || rows >= batch || dstCapFloats < (int64_t)c->T * d ) return WH_E_BOUNDS;
		const f16* base = ( w == "cross-k1" ? c->crossK : c->crossV ) + ( (int64_t)layer * c->maxBatch + rows ) * c->T * d;
		return readHeadMajor( c, base, 1, c->T, c->T, dstHost );
	}
	if( w == "encode-out" )
	{
		// the FP16 LayerNorm output that feeds the cross-attention projection (of a batch the encoder took in one chunk)
		if( batch > c->encChunk ) { setError( "debug_read: encode-out holds the last encoder chunk only" ); return WH_E_BOUNDS; }
		const int64_t n = (int64_t)batch * c->T * d;
		if( dstCapFloats < n ) return WH_E_BOUNDS;
		std::vector<uint16_t> tmp( (size_t)n );
		WH_HIP( hipStreamSynchronize( c->stream ) );
		WH_HIP( hipMemcpy( tmp.data(), c->xn, (size_t)n * 2, hipMemcpyDeviceToHost ) );
		for( int64_t i = 0; i < n; i++ ) dstHost[ i ] = f16BitsToF32( tmp[ (size_t)i ] );
		return 0;
	}
	auto readF16 = [ & ]( const f16* src, int64_t n ) -> int
	{
		if( !src ) { setError( "debug_read: nothing captured (set WH_FLAG_DEBUG_CAPTURE before the call that produces it)" ); return WH_E_NOT_READY; }
		if( dstCapFloats < n ) return WH_E_BOUNDS;
		std::vector<uint16_t> tmp( (size_t)n );
		WH_HIP( hipStreamSynchronize( c->stream ) );
		WH_HIP( hipMemcpy( tmp.data(), src, (size_t)n * 2, hipMemcpyDeviceToHost ) );
		for( int64_t i = 0; i < n; i++ ) dstHost[ i ] = f16BitsToF32( tmp[ (size_t)i ] );
		return 0;
	};
	if( w == "enc.temp1" )
	{
		// conv1 + bias + GELU, time-major [batch][2*n_ctx][d] (the reference's tensor is [d][2*n_ctx]); padding rows dropped
		if( !c->capTemp1 ) return readF16( nullptr, 0 );
		const int64_t rowsT = 2ll * c->T;
		if( dstCapFloats < batch * rowsT * d ) return WH_E_BOUNDS;
		std::vector<uint16_t> tmp( (size_t)( batch * c->conv1Stride ) );
		WH_HIP( hipStreamSynchronize( c->stream ) );
		WH_HIP( hipMemcpy( tmp.data(), c->capTemp1, tmp.size() * 2, hipMemcpyDeviceToHost ) );
		for( int b = 0; b < batch; b++ )
			for( int64_t i = 0; i < rowsT * d; i++ ) dstHost[ b * rowsT * d + i ] = f16BitsToF32( tmp[ (size_t)( b * c->conv1Stride + d + i ) ] );
		return 0;
	}
	if( w == "enc.layer0.in" )
	{
		const int64_t n = (int64_t)batch * c->T * d;
		if( !c->capLayer0In ) return readF16( nullptr, 0 );
		if( dstCapFloats < n ) return WH_E_BOUNDS;
		WH_HIP( hipStreamSynchronize( c->stream ) );
		WH_HIP( hipMemcpy( dstHost, c->capLayer0In, (size_t)n * 4, hipMemcpyDeviceToHost ) );
		return 0;
	}
	if( w == "enc-KQV" ) return readF16( c->capEncKqv, (int64_t)batch * c->T * d );	   // layer 0, [batch][n_ctx][d] (heads side by side)
	if( w == "dec-KQV" ) return readF16( c->capDecKqvSelf, (int64_t)c->capDecRows * d );   // layer 0 self-attention output, rows of the last wh_decode
	if( w == "dec-KQV#2" ) return readF16( c->capDecKqvCross, (int64_t)c->capDecRows * d );
	if( layer < 0 || layer >= hp.n_text_layer ) return WH_E_BOUNDS;
	if( w == "cross-k" || w == "cross-v" )
	{
		if( dstCapFloats < (int64_t)batch * c->T * d ) return WH_E_BOUNDS;
		const f16* base = ( w == "cross-k" ? c->crossK : c->crossV ) + (int64_t)layer * c->maxBatch * c->T * d;
		return readHeadMajor( c, base, batch, c->T, c->T, dstHost );
	}
	if( w == "self-k" || w == "self-v" )
	{
		if( rows <= 0 || rows > hp.n_text_ctx || dstCapFloats < (int64_t)seqs * rows * d ) return WH_E_BOUNDS;
		const f16* base = ( w == "self-k" ? c->selfK : c->selfV ) + (int64_t)layer * c->maxSeq * hp.n_text_ctx * d;
		return readHeadMajor( c, base, seqs, rows, hp.n_text_ctx, dstHost );
	}
	setError( "debug_read: unknown item" );
	return WH_E_INVALIDARG;
}

// ==================================================================================================================
// op-level entry points
// ==================================================================================================================
int wh_op_mul_mat( void* stream, const void* aF16, const void* wF16, const float* bias, const float* residual, float* out, int M, int N, int K )
{
	GemmArgs g = plainGemm( (const f16*)aF16, (const f16*)wF16, M, N, K );
	g.epi = EPI_F32; g.bias = bias; g.res = residual; g.out32 = out;
	if( M > 32 && M <= GEMV_FUSED_MAX_ROWS && N <= 2048 && K >= 2048 )
	{
		// option dec_split needs room for its eight partial tiles: no context here, so the op-level entry keeps one buffer per device (calls on different streams
		// would share it: this entry point is the tests' and tools', the decoder passes its context's buffer)
		static std::mutex mx;
		static float* bufs[ 64 ] = {};
		int dev = 0;
		WH_HIP( hipGetDevice( &dev ) );
		std::lock_guard<std::mutex> lk( mx );
		if( !bufs[ dev & 63 ] ) WH_HIP( hipMalloc( (void**)&bufs[ dev & 63 ], 8ull * GEMV_FUSED_MAX_ROWS * 2048 * sizeof( float ) ) );
		g.splitScratch = bufs[ dev & 63 ];
	}
	// the same choice the decoder makes: up to 32 rows go to the gemv when K allows it
	if( M <= GEMV_MAX_ROWS && ( K % 128 ) == 0 ) return launchGemv( g, (hipStream_t)stream );	   // the op-level entry is the decode-step product: up to 512 rows on the weight-streaming kernels
	return M <= 32 ? launchGemmSkinny( g, (hipStream_t)stream ) : launchGemm( g, (hipStream_t)stream );
}

int wh_op_mul_mat_gelu( void* stream, const void* aF16, const void* wF16, const float* bias, void* outF16, int M, int N, int K )
{
	if( !bias ) { setError( "mul_mat_gelu: bias is required" ); return WH_E_INVALIDARG; }
	GemmArgs g = plainGemm( (const f16*)aF16, (const f16*)wF16, M, N, K );
	g.epi = EPI_F16_GELU; g.bias = bias; g.out16 = (f16*)outF16;
	if( M <= GEMV_MAX_ROWS && ( K % 128 ) == 0 ) return launchGemv( g, (hipStream_t)stream );	   // the op-level entry is the decode-step product: up to 512 rows on the weight-streaming kernels
	return M <= 32 ? launchGemmSkinny( g, (hipStream_t)stream ) : launchGemm( g, (hipStream_t)stream );
}

int wh_op_layer_norm( void* stream, const float* x, const float* w, const float* b, void* outF16, int rows, int d )
{
	return launchLayerNorm( x, w, b, (f16*)outF16, rows, d, (hipStream_t)stream );
}

int wh_op_flash_attention( void* stream, const void* q, const void* k, const void* vT, void* out, int batch, int heads, int nCtx )
{
	// no model here: the op-level entry keeps its own copy of the exponential's table per device (built like wh_model_finalize builds the arena's)
	const f16* expTab = nullptr;
	if( g_tuning & TUNE_ATTN_ENC_TABLE )
	{
		static std::mutex mx;
		static f16* tabs[ 64 ] = {};
		int dev = 0;
		WH_HIP( hipGetDevice( &dev ) );
		std::lock_guard<std::mutex> lk( mx );
		if( !tabs[ dev & 63 ] )
		{
			std::vector<_Float16> tab( EXP_TABLE_ENTRIES );
			for( uint32_t i = 0; i < (uint32_t)EXP_TABLE_ENTRIES; i++ )
			{
				const uint16_t bits = (uint16_t)( i | 0x8000u );
				_Float16 h;
				memcpy( &h, &bits, 2 );
				tab[ i ] = (_Float16)expf( (float)h );
			}
			WH_HIP( hipMalloc( (void**)&tabs[ dev & 63 ], EXP_TABLE_ENTRIES * 2 ) );
			WH_HIP( hipMemcpy( tabs[ dev & 63 ], tab.data(), EXP_TABLE_ENTRIES * 2, hipMemcpyHostToDevice ) );
		}
		expTab = tabs[ dev & 63 ];
	}
	return launchAttentionEnc( (const f16*)q, (const f16*)k, (const f16*)vT, (f16*)out, batch, heads, nCtx, roundUp( nCtx, 256 ), false, expTab, (hipStream_t)stream );
}

int wh_op_decoder_attention( void* stream, const void* qF16, const void* kCache, const void* vCache, void* outF16, int sequences, int heads,
	int nTok, int nKeys, int keyStride, int causal, int nPast, int group, int parityThreads )
{
	DecAttnArgs a = {};
	a.q = (const f16*)qF16; a.kc = (const f16*)kCache; a.vc = (const f16*)vCache; a.out = (f16*)outF16;
	a.batch = sequences; a.H = heads; a.nTok = nTok; a.nKeys = nKeys; a.keyStride = keyStride;
	a.causal = causal; a.nPast = nPast; a.group = group; a.parityThreads = parityThreads;
	return launchAttentionDec( a, (hipStream_t)stream );
}

int wh_op_decoder_cross_attention( void* stream, const float* x, const float* lnW, const float* lnB, const void* qW, const float* qB, float qScale,
	const void* kCache, const void* vCache, void* outF16, int sequences, int heads, int nKeys, int keyStride, int group )
{
	if( !x || !lnW || !lnB || !qW || !qB ) { setError( "decoder_cross_attention: null argument" ); return WH_E_INVALIDARG; }
	DecAttnArgs a = {};
	a.kc = (const f16*)kCache; a.vc = (const f16*)vCache; a.out = (f16*)outF16;
	a.batch = sequences; a.H = heads; a.nTok = 1; a.nKeys = nKeys; a.keyStride = keyStride;
	a.group = group;
	a.lnX = x; a.lnW = lnW; a.lnB = lnB; a.qW = (const f16*)qW; a.qB = qB; a.qScale = qScale;
	return launchAttentionDec( a, (hipStream_t)stream );
}

int wh_op_soft_max( void* stream, float* x, int rows, int cols )
{
	return launchSoftMaxRows( x, rows, cols, (hipStream_t)stream );
}


// ------------------------------------------------------------------------------------------------------------------
// micro-benchmarks used by tools/gemm_probe.py (development aid; not on the product path)
// ------------------------------------------------------------------------------------------------------------------
namespace
{
	// Grid-wide barrier cost probe: `n` rounds of { every workgroup publishes 64 bytes, barrier, reads another workgroup's
	// line and checks it }. One monotonic counter, agent-scope release on arrival, acquire polling, bounded spin.
	// mode 0: every workgroup polls the one counter; mode 1: arrivals are counted per XCD (workgroup id % 8) and the last
	// arrival of an XCD bumps the global counter by its XCD's population, so only 8 RMWs hit the shared line.
	__global__ void __launch_bounds__( 256 ) probeGridBarrier( unsigned* counters, float* buf, int n, int mode, int* err )
	{
		const unsigned G = gridDim.x, wg = blockIdx.x;
		const int lane = threadIdx.x;
		unsigned* const global = counters;
		unsigned* const perXcd = counters + 64 + ( wg & 7 ) * 64;	// own 256-byte line per counter
		const unsigned xcdPop = ( G + 7 - ( wg & 7 ) ) / 8;
		__shared__ int dead;
		if( lane == 0 ) dead = 0;
		for( int it = 0; it < n; it++ )
		{
			float* const slot = buf + ( it & 1 ) * G * 16;
			if( lane < 16 ) slot[ wg * 16 + lane ] = (float)( it * 7 + (int)wg + lane );
			__syncthreads();
			if( lane == 0 )
			{
				const unsigned target = (unsigned)( it + 1 ) * G;
				if( mode == 0 )
					__hip_atomic_fetch_add( global, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT );
				else
				{
					const unsigned prev = __hip_atomic_fetch_add( perXcd, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT );
					if( ( prev + 1 ) % xcdPop == 0 )
						__hip_atomic_fetch_add( global, xcdPop, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT );
				}
				unsigned spins = 0;
				while( __hip_atomic_load( global, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT ) < target )
				{
					if( ++spins > ( 1u << 21 ) )
					{
						*err = 2;
						dead = 1;
						break;
					}
				}
			}
			__syncthreads();
			if( dead ) return;
			const unsigned other = ( wg + 97 ) % G;
			if( lane < 16 )
			{
				const float got = __builtin_nontemporal_load( slot + other * 16 + lane );
				if( got != (float)( it * 7 + (int)other + lane ) ) *err = 1;
			}
		}
	}
	// max |a - b| over n floats (non-negative floats order like their bit patterns; NaN counts as +inf)
	__global__ void probeMaxDiff( const float* a, const float* b, long long n, int* out )
	{
		float m = 0.0f;
		for( long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x )
		{
			const float d = fabsf( a[ i ] - b[ i ] );
			m = ( d != d ) ? INFINITY : fmaxf( m, d );
		}
		for( int off = 32; off > 0; off >>= 1 ) m = fmaxf( m, __shfl_xor( m, off ) );
		if( ( threadIdx.x & 63 ) == 0 ) atomicMax( out, __float_as_int( m ) );
	}
	__global__ void probeFill( _Float16* p, long long n, unsigned seed )
	{
		for( long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x )
		{
			unsigned h = (unsigned)i * 2654435761u + seed;
			h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
			p[ i ] = (_Float16)( ( (float)( h & 0xFFFF ) / 32768.0f - 1.0f ) * 0.5f );
		}
	}
}

int wh_debug_set_tuning( uint32_t mask )
{
	g_tuning = mask;
	return 0;
}

int wh_debug_set_option( const char* name, int value )
{
	if( !name ) { setError( "debug_set_option: null name" ); return WH_E_INVALIDARG; }
	for( const OptionName& o : g_optionNames )
		if( 0 == strcmp( o.name, name ) )
		{
			if( value < -1 || value > ( 1 << 24 ) ) { setError( std::string( "option '" ) + name + "': value out of range" ); return WH_E_INVALIDARG; }
			g_opt.*( o.field ) = value;
			return 0;
		}
	setError( "debug_set_option: unknown option" );
	return WH_E_INVALIDARG;
}

int wh_debug_get_option( const char* name, int* value )
{
	if( !name || !value ) { setError( "debug_get_option: null argument" ); return WH_E_INVALIDARG; }
	for( const OptionName& o : g_optionNames )
		if( 0 == strcmp( o.name, name ) )
		{
			*value = g_opt.*( o.field );
			return 0;
		}
	setError( "debug_get_option: unknown option" );
	return WH_E_INVALIDARG;
}

int wh_debug_probe( wh_context* c, int kind, int variant, int M, int N, int K, int iters, float* msPerIter )
{
	if( !c || !msPerIter || iters <= 0 ) return WH_E_INVALIDARG;
	WH_BIND( c->m );
	hipStream_t st = c->stream;
	hipEvent_t e0, e1;
	WH_HIP( hipEventCreate( &e0 ) );
	WH_HIP( hipEventCreate( &e1 ) );
	int rc = 0;
	float ms = 0;
	if( kind == 0 || kind == 2 )
	{
		// chain of dependent empty kernels, `variant` workgroups of 256 threads each; kind 2 replays them from a graph
		const int grid = variant > 0 ? variant : 1;
		hipGraphExec_t exec = nullptr;
		const int perGraph = 100;
		if( kind == 2 )
		{
			hipGraph_t g = nullptr;
			WH_HIP( hipStreamBeginCapture( st, hipStreamCaptureModeThreadLocal ) );
			for( int i = 0; i < perGraph; i++ ) hipLaunchKernelGGL( probeEmpty, dim3( grid ), dim3( 256 ), 0, st, (int*)nullptr );
			WH_HIP( hipStreamEndCapture( st, &g ) );
			WH_HIP( hipGraphInstantiate( &exec, g, nullptr, nullptr, 0 ) );
			(void)hipGraphDestroy( g );
			WH_HIP( hipGraphLaunch( exec, st ) );
		}
		for( int i = 0; i < 20; i++ ) hipLaunchKernelGGL( probeEmpty, dim3( grid ), dim3( 256 ), 0, st, (int*)nullptr );
		WH_HIP( hipStreamSynchronize( st ) );
		WH_HIP( hipEventRecord( e0, st ) );
		if( kind == 2 )
			for( int i = 0; i < ( iters + perGraph - 1 ) / perGraph; i++ ) WH_HIP( hipGraphLaunch( exec, st ) );
		else
			for( int i = 0; i < iters; i++ ) hipLaunchKernelGGL( probeEmpty, dim3( grid ), dim3( 256 ), 0, st, (int*)nullptr );
		WH_HIP( hipEventRecord( e1, st ) );
		WH_HIP( hipEventSynchronize( e1 ) );
		WH_HIP( hipEventElapsedTime( &ms, e0, e1 ) );
		if( kind == 2 ) ms = ms * (float)iters / (float)( ( iters + perGraph - 1 ) / perGraph * perGraph );
		if( exec ) (void)hipGraphExecDestroy( exec );
	}
	else if( kind == 1 )
	{
		void *A = nullptr, *W = nullptr, *out = nullptr;
		WH_HIP( hipMalloc( &A, (size_t)M * K * 2 ) );
		WH_HIP( hipMalloc( &W, (size_t)N * K * 2 ) );
		WH_HIP( hipMalloc( &out, (size_t)M * N * 4 ) );
		hipLaunchKernelGGL( probeFill, dim3( 1024 ), dim3( 256 ), 0, st, (_Float16*)A, (long long)M * K, 1u );
		hipLaunchKernelGGL( probeFill, dim3( 1024 ), dim3( 256 ), 0, st, (_Float16*)W, (long long)N * K, 2u );
		GemmArgs g = plainGemm( (const f16*)A, (const f16*)W, M, N, K );
		g.epi = EPI_F32; g.out32 = (float*)out;
		g.groupM = variant / 100;	   // variant = tile variant + 100 * M tiles per band of the block walk (0 = default)
		variant %= 100;
		unsigned long long* stamps = nullptr;
		if( variant == 35 || variant == 33 )
		{
			WH_HIP( hipMalloc( (void**)&stamps, 32 ) );
			WH_HIP( hipMemsetAsync( stamps, 0, 32, st ) );
			g.pe = (const float*)stamps;
		}
		for( int i = 0; i < 2 && rc == 0; i++ ) rc = launchGemmVariant( g, variant, st );
		if( rc == 0 )
		{
			WH_HIP( hipStreamSynchronize( st ) );
			WH_HIP( hipEventRecord( e0, st ) );
			for( int i = 0; i < iters && rc == 0; i++ ) rc = launchGemmVariant( g, variant, st );
			WH_HIP( hipEventRecord( e1, st ) );
			WH_HIP( hipEventSynchronize( e1 ) );
			WH_HIP( hipEventElapsedTime( &ms, e0, e1 ) );
		}
		if( stamps && rc == 0 )
		{
			unsigned long long h[ 4 ] = {};
			WH_HIP( hipMemcpy( h, stamps, 32, hipMemcpyDeviceToHost ) );
			const double n = h[ 3 ] ? (double)h[ 3 ] : 1.0;
			fprintf( stderr, "[gemm8 stamps] %dx%dx%d: per tile (wave 0, cycles) first-operand wait %.0f, K loop %.0f, epilogue %.0f; %.0f tiles\n", M, N, K,
				h[ 0 ] / n, h[ 1 ] / n, h[ 2 ] / n, n );
		}
		if( stamps ) (void)hipFree( stamps );
		// every variant is checked against the production path on the same operands: a pipeline that races is fast and wrong
		// (variants 31 .. 39 are ablations, wrong by construction)
		if( rc == 0 && ( variant == 34 || ( !( variant >= 31 && variant <= 39 ) && !( variant >= 41 && variant <= 49 ) && !( variant >= 60 && variant <= 69 ) ) ) )
		{
			void* ref = nullptr;
			int* diff = nullptr;
			WH_HIP( hipMalloc( &ref, (size_t)M * N * 4 ) );
			WH_HIP( hipMalloc( (void**)&diff, 4 ) );
			WH_HIP( hipMemsetAsync( diff, 0, 4, st ) );
			GemmArgs g2 = plainGemm( (const f16*)A, (const f16*)W, M, N, K );
			g2.epi = EPI_F32; g2.out32 = (float*)ref;
			static const int refVariant = []() { const char* e = getenv( "WH_PROBE_REF" ); return e ? atoi( e ) : 2; }();
			rc = launchGemmVariant( g2, refVariant, st );	   // 2 = the register-staged 128x128x32 kernel: shares no staging or scheduling code with the variants under test
			if( rc == 0 )
			{
				hipLaunchKernelGGL( probeMaxDiff, dim3( 2048 ), dim3( 256 ), 0, st, (const float*)out, (const float*)ref, (long long)M * N, diff );
				int bits = 0;
				WH_HIP( hipMemcpyAsync( &bits, diff, 4, hipMemcpyDeviceToHost, st ) );
				WH_HIP( hipStreamSynchronize( st ) );
				float md;
				memcpy( &md, &bits, 4 );
				if( getenv( "WH_PROBE_REF" ) ) fprintf( stderr, "[gemm probe] variant %d against variant %d, %d x %d x %d: max |diff| = %g\n", variant, refVariant, M, N, K, (double)md );
				if( !( md <= 1e-3f ) )
				{
					char buf[ 160 ];
					snprintf( buf, sizeof( buf ), "gemm probe: variant %d differs from the production kernel by %g", variant, (double)md );
					setError( buf );
					rc = -1;
				}
			}
			(void)hipFree( ref ); (void)hipFree( diff );
		}
		(void)hipFree( A ); (void)hipFree( W ); (void)hipFree( out );
	}
	else if( kind == 3 )
	{
		// grid barrier: `variant` workgroups (must all be resident: <= number of CUs), M = mode, iters barriers per launch
		const int grid = variant > 0 ? variant : 256;
		unsigned* counters = nullptr;
		float* buf = nullptr;
		int* err = nullptr;
		WH_HIP( hipMalloc( &counters, 4096 ) );
		WH_HIP( hipMalloc( &buf, (size_t)grid * 16 * 2 * 4 ) );
		WH_HIP( hipMalloc( &err, 4 ) );
		WH_HIP( hipMemsetAsync( err, 0, 4, st ) );
		for( int rep = 0; rep < 2; rep++ )
		{
			WH_HIP( hipMemsetAsync( counters, 0, 4096, st ) );
			WH_HIP( hipStreamSynchronize( st ) );
			WH_HIP( hipEventRecord( e0, st ) );
			hipLaunchKernelGGL( probeGridBarrier, dim3( grid ), dim3( 256 ), 0, st, counters, buf, iters, M, err );
			WH_HIP( hipEventRecord( e1, st ) );
			WH_HIP( hipEventSynchronize( e1 ) );
			WH_HIP( hipEventElapsedTime( &ms, e0, e1 ) );
		}
		int herr = 0;
		WH_HIP( hipMemcpy( &herr, err, 4, hipMemcpyDeviceToHost ) );
		if( herr != 0 )
		{
			setError( herr == 1 ? "grid barrier probe: stale data after the barrier" : "grid barrier probe: spin limit reached" );
			rc = -1;
		}
		(void)hipFree( counters ); (void)hipFree( buf ); (void)hipFree( err );
	}
	else if( kind == 4 )
	{
		// MFMA-bound product NEXT TO the HBM-bound cross-attention, measured directly: the encoder product of M x N x K (the production
		// launch: persistent gemmTiled8) `iters` times on stream A, attentionDecG<1,...> over `variant` windows x 16 heads x 1500 keys on stream B as often as fills
		// the same time; each alone, then together -- for the persistent grid limited to 256 / 224 / 192 / 160 / 128 CUs, without CU masks (the attention's
		// workgroups land where LDS and registers are free) and with complementary CU masks on the two streams. Results on stderr; ms = the unmasked 256-CU pair.
		const int wins = variant > 0 ? variant : 224, H = 16, T = 1500, d = H * HEAD_DIM;
		void *A = nullptr, *W = nullptr, *out = nullptr, *kc = nullptr, *vc = nullptr, *q = nullptr, *ao = nullptr;
		WH_HIP( hipMalloc( &A, (size_t)M * K * 2 ) );
		WH_HIP( hipMalloc( &W, (size_t)N * K * 2 ) );
		WH_HIP( hipMalloc( &out, (size_t)M * N * 4 ) );
		const size_t kvBytes = (size_t)wins * H * T * HEAD_DIM * 2;
		WH_HIP( hipMalloc( &kc, kvBytes ) );
		WH_HIP( hipMalloc( &vc, kvBytes ) );
		WH_HIP( hipMalloc( &q, (size_t)wins * d * 2 ) );
		WH_HIP( hipMalloc( &ao, (size_t)wins * d * 2 ) );
		hipLaunchKernelGGL( probeFill, dim3( 1024 ), dim3( 256 ), 0, st, (_Float16*)A, (long long)M * K, 1u );
		hipLaunchKernelGGL( probeFill, dim3( 1024 ), dim3( 256 ), 0, st, (_Float16*)W, (long long)N * K, 2u );
		hipLaunchKernelGGL( probeFill, dim3( 1024 ), dim3( 256 ), 0, st, (_Float16*)kc, (long long)( kvBytes / 2 ), 3u );
		hipLaunchKernelGGL( probeFill, dim3( 1024 ), dim3( 256 ), 0, st, (_Float16*)vc, (long long)( kvBytes / 2 ), 4u );
		hipLaunchKernelGGL( probeFill, dim3( 1024 ), dim3( 256 ), 0, st, (_Float16*)q, (long long)wins * d, 5u );
		WH_HIP( hipStreamSynchronize( st ) );
		GemmArgs g = plainGemm( (const f16*)A, (const f16*)W, M, N, K );
		g.epi = EPI_F32; g.out32 = (float*)out;
		DecAttnArgs a = {};
		a.q = (const f16*)q; a.kc = (const f16*)kc; a.vc = (const f16*)vc; a.out = (f16*)ao;
		a.batch = wins; a.H = H; a.nTok = 1; a.nKeys = T; a.keyStride = T; a.causal = 0; a.group = 1;
		int cus = 0;
		WH_HIP( hipDeviceGetAttribute( &cus, hipDeviceAttributeMultiprocessorCount, c->m->device ) );
		auto wall = []() { return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); };
		auto timeIt = [ & ]( hipStream_t sa, hipStream_t sb, int nA, int nB, int cuLimit, double& msOut ) -> int
		{
			GemmArgs gl = g;
			gl.cuLimit = cuLimit;
			WH_HIP( hipDeviceSynchronize() );
			const double t0 = wall();
			// interleave the enqueues so that neither stream starts far ahead of the other
			const int n = std::max( nA, nB );
			for( int i = 0; i < n; i++ )
			{
				if( i < nA ) WH_CHECK( launchGemm( gl, sa ) );
				if( i < nB ) WH_CHECK( launchAttentionDec( a, sb ) );
			}
			WH_HIP( hipDeviceSynchronize() );
			msOut = wall() - t0;
			return 0;
		};
		hipStream_t sa = nullptr, sb = nullptr;
		WH_HIP( hipStreamCreateWithFlags( &sa, hipStreamNonBlocking ) );
		WH_HIP( hipStreamCreateWithFlags( &sb, hipStreamNonBlocking ) );
		double tA1 = 0, tB1 = 0, dummy = 0;
		rc = timeIt( sa, sb, 3, 3, 0, dummy );	   // warm-up
		if( rc == 0 ) rc = timeIt( sa, sb, iters, 0, 0, tA1 );
		if( rc == 0 ) rc = timeIt( sa, sb, 0, iters, 0, tB1 );
		const int nA = iters, nB = std::max( 1, (int)( iters * tA1 / std::max( tB1, 1e-6 ) + 0.5 ) );
		double tA = 0, tB = 0;
		if( rc == 0 ) rc = timeIt( sa, sb, nA, 0, 0, tA );
		if( rc == 0 ) rc = timeIt( sa, sb, 0, nB, 0, tB );
		if( rc == 0 )
		{
			fprintf( stderr, "[pair] product %d x %d x %d: %d launches %.2f ms alone (%.0f TFLOP/s); cross-attention %d windows: %d launches %.2f ms alone (%.2f TB/s)\n", M, N, K, nA, tA,
				2.0 * M * N * K * nA / tA * 1e-9, wins, nB, tB, 2.0 * kvBytes * nB / tB * 1e-9 );
			const int limits[] = { 0, 224, 192, 160, 128 };
			for( int li = 0; li < 5 && rc == 0; li++ )
			{
				const int lim = limits[ li ];
				double tAl = 0, tp = 0;
				rc = timeIt( sa, sb, nA, 0, lim, tAl );
				if( rc == 0 ) rc = timeIt( sa, sb, nA, nB, lim, tp );
				if( rc == 0 ) fprintf( stderr, "[pair] no masks, product on %3d CUs: product alone %.2f ms, pair %.2f ms = %.3f of (A + B) = %.3f of (A at this limit + B)\n", lim ? lim : cus, tAl, tp,
					tp / ( tA + tB ), tp / ( tAl + tB ) );
				if( li == 0 ) ms = (float)tp * (float)iters;
			}
			// complementary CU masks: CU i of the mask belongs to XCD i % 8 (a multiple of 8 keeps the split even over the XCDs)
			for( int li = 1; li < 5 && rc == 0; li++ )
			{
				const int lim = limits[ li ];
				uint32_t mA[ 16 ] = {}, mB[ 16 ] = {};
				const uint32_t words = (uint32_t)( ( cus + 31 ) / 32 );
				for( int i = 0; i < cus && i < 512; i++ ) ( i < lim ? mA : mB )[ i >> 5 ] |= 1u << ( i & 31 );
				hipStream_t ma = nullptr, mb = nullptr;
				if( hipExtStreamCreateWithCUMask( &ma, words, mA ) != hipSuccess || hipExtStreamCreateWithCUMask( &mb, words, mB ) != hipSuccess ) { fprintf( stderr, "[pair] CU-masked streams unavailable\n" ); break; }
				double tAl = 0, tBl = 0, tp = 0;
				rc = timeIt( ma, mb, nA, 0, lim, tAl );
				if( rc == 0 ) rc = timeIt( ma, mb, 0, nB, lim, tBl );
				if( rc == 0 ) rc = timeIt( ma, mb, nA, nB, lim, tp );
				if( rc == 0 ) fprintf( stderr, "[pair] masks %3d | %3d CUs: product alone %.2f ms, attention alone %.2f ms (%.2f TB/s), pair %.2f ms = %.3f of (A + B at 256 CUs)\n", lim, cus - lim, tAl, tBl,
					2.0 * kvBytes * nB / tBl * 1e-9, tp, tp / ( tA + tB ) );
				(void)hipStreamDestroy( ma ); (void)hipStreamDestroy( mb );
			}
		}
		(void)hipStreamDestroy( sa ); (void)hipStreamDestroy( sb );
		(void)hipFree( A ); (void)hipFree( W ); (void)hipFree( out ); (void)hipFree( kc ); (void)hipFree( vc ); (void)hipFree( q ); (void)hipFree( ao );
	}
	else
		rc = WH_E_INVALIDARG;
	(void)hipEventDestroy( e0 );
	(void)hipEventDestroy( e1 );
	*msPerIter = ms / (float)iters;
	return rc;
}

}	// extern "C"
