// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY. Never linked into or called by the product path.
//
// Builds the reference's own CPU implementation (Whisper/source/whisper.cpp + ggml.c, the vendored
// whisper.cpp that Const-me/Whisper exposes as eModelImplementation::Reference, Whisper/whisperCom.cpp:52)
// UNMODIFIED, from where it lies under /root/reference, into oracle/_ref/libwhisper_ref.so, and adds a flat
// C API around it so Python tests can (a) run encode/decode/full, (b) read logits, KV caches and the named
// intermediate tensors the reference itself marks as probe points (Tracing::delayTensor, whisper.cpp:1121-1869).
//
// Like whisperCom.cpp:52 we #include the .cpp so the file-static whisper_context internals are reachable.
// No reference source is copied into this repository; the include path is given by oracle/Makefile.
#include "whisper.cpp"

#include <cstdarg>
#include <mutex>

// ---- logger shim definitions (declared in shim/Utils/Logger.h) ----
static int g_logLevel = 1;	// 0 = silent, 1 = errors, 2 = everything
static void vlog( int lvl, const char* fmt, va_list args )
{
	if( lvl > g_logLevel ) return;
	vfprintf( stderr, fmt, args );
	fputc( '\n', stderr );
}
extern "C" {
void logError( const char8_t* fmt, ... ) { va_list a; va_start( a, fmt ); vlog( 1, (const char*)fmt, a ); va_end( a ); }
void logWarning( const char8_t* fmt, ... ) { va_list a; va_start( a, fmt ); vlog( 2, (const char*)fmt, a ); va_end( a ); }
void logInfo( const char8_t* fmt, ... ) { va_list a; va_start( a, fmt ); vlog( 2, (const char*)fmt, a ); va_end( a ); }
void logDebug( const char8_t* fmt, ... ) { va_list a; va_start( a, fmt ); vlog( 3, (const char*)fmt, a ); va_end( a ); }
}

// ---- tracing shim definitions (declared in shim/trace_shim.h) ----
namespace Tracing
{
	bool g_enabled = false;
	std::map<std::string, Captured> g_captured;
	static std::vector<std::pair<std::string, const ggml_tensor*>> g_delayed;

	void captureTensor( const char* name, const ggml_tensor* t )
	{
		Captured& c = g_captured[ name ];
		for( int i = 0; i < 4; i++ ) c.ne[ i ] = t->ne[ i ];
		const size_t n = (size_t)t->ne[ 0 ] * t->ne[ 1 ] * t->ne[ 2 ] * t->ne[ 3 ];
		c.data.resize( n );
		size_t o = 0;
		for( int i3 = 0; i3 < t->ne[ 3 ]; i3++ )
			for( int i2 = 0; i2 < t->ne[ 2 ]; i2++ )
				for( int i1 = 0; i1 < t->ne[ 1 ]; i1++ )
					for( int i0 = 0; i0 < t->ne[ 0 ]; i0++ )
					{
						const char* p = (const char*)t->data + i0 * t->nb[ 0 ] + i1 * t->nb[ 1 ] + i2 * t->nb[ 2 ] + i3 * t->nb[ 3 ];
						float v;
						if( t->type == GGML_TYPE_F32 ) v = *(const float*)p;
						else if( t->type == GGML_TYPE_F16 ) v = ggml_fp16_to_fp32( *(const ggml_fp16_t*)p );
						else if( t->type == GGML_TYPE_I32 ) v = (float)*(const int32_t*)p;
						else v = 0;
						c.data[ o++ ] = v;
					}
	}
	void delayTensor( const ItemName& name, const ggml_tensor* t )
	{
		if( !g_enabled ) return;
		// the decoder uses "dec-KQV" for both the self- and the cross-attention product of layer 0
		std::string key = name.text;
		for( const auto& d : g_delayed )
			if( d.first == key ) { key += "#2"; break; }
		g_delayed.emplace_back( key, t );
	}
	void writeDelayedTensors()
	{
		if( g_enabled )
			for( const auto& d : g_delayed )
				captureTensor( d.first.c_str(), d.second );
		g_delayed.clear();
	}
}

static void copyF16( const ggml_tensor* t, size_t offsetElts, size_t count, float* dst )
{
	const ggml_fp16_t* src = (const ggml_fp16_t*)t->data + offsetElts;
	for( size_t i = 0; i < count; i++ ) dst[ i ] = ggml_fp16_to_fp32( src[ i ] );
}

extern "C" {

void ref_set_log_level( int lvl ) { g_logLevel = lvl; }

whisper_context* ref_init( const char* path )
{
	whisper_context* ctx = whisper_init( path );
	// whisper_context::exp_n_audio_ctx has no initialiser (whisper.cpp:431) and is only set by whisper_full
	// (whisper.cpp:2813); direct whisper_encode / whisper_decode calls would otherwise read garbage.
	if( ctx ) ctx->exp_n_audio_ctx = 0;
	return ctx;
}
void ref_set_audio_ctx( whisper_context* ctx, int n ) { ctx->exp_n_audio_ctx = n; }
void ref_free( whisper_context* ctx ) { whisper_free( ctx ); }

// hparams in file order: n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer, n_text_ctx, n_text_state,
// n_text_head, n_text_layer, n_mels, f16
void ref_hparams( whisper_context* ctx, int32_t* out11 )
{
	const auto& h = ctx->model.hparams;
	const int32_t v[ 11 ] = { h.n_vocab, h.n_audio_ctx, h.n_audio_state, h.n_audio_head, h.n_audio_layer,
		h.n_text_ctx, h.n_text_state, h.n_text_head, h.n_text_layer, h.n_mels, h.f16 };
	memcpy( out11, v, sizeof( v ) );
}

int ref_pcm_to_mel( whisper_context* ctx, const float* pcm, int n, int nThreads ) { return whisper_pcm_to_mel( ctx, pcm, n, nThreads ); }
int ref_set_mel( whisper_context* ctx, const float* mel, int nLen, int nMel ) { return whisper_set_mel( ctx, mel, nLen, nMel ); }
// whisper_set_mel refuses anything but WHISPER_N_MEL = 80 bins (whisper.cpp:2318-2321) although the encoder itself takes the count from the
// model file (whisper.cpp:1097-1106): writing the context's spectrogram directly lets the reference's encoder / decoder run a model of the
// large-v3 SHAPE (128 mel bins), which no entry point of the reference can feed
int ref_set_mel_any( whisper_context* ctx, const float* mel, int nLen, int nMel )
{
	ctx->mel.n_len = nLen;
	ctx->mel.n_mel = nMel;
	ctx->mel.data.assign( mel, mel + (size_t)nLen * nMel );
	return 0;
}
int ref_mel_len( whisper_context* ctx ) { return ctx->mel.n_len; }
void ref_get_mel( whisper_context* ctx, float* dst ) { memcpy( dst, ctx->mel.data.data(), ctx->mel.data.size() * sizeof( float ) ); }

int ref_encode( whisper_context* ctx, int melOffset, int nThreads ) { return whisper_encode( ctx, melOffset, nThreads ); }
int ref_decode( whisper_context* ctx, const int32_t* tokens, int nTokens, int nPast, int nThreads )
{
	return whisper_decode( ctx, tokens, nTokens, nPast, nThreads );
}
// logits / probs of the last decode call, [nTokens][n_vocab] (whisper.cpp:1855-1859)
size_t ref_logits_size( whisper_context* ctx ) { return ctx->logits.size(); }
void ref_get_logits( whisper_context* ctx, float* dst ) { memcpy( dst, ctx->logits.data(), ctx->logits.size() * sizeof( float ) ); }
void ref_get_probs( whisper_context* ctx, float* dst ) { memcpy( dst, ctx->probs.data(), ctx->probs.size() * sizeof( float ) ); }

// cross-attention cache of one decoder layer, token-major [n_audio_ctx][n_state] (whisper.cpp:1479-1483), FP16 -> FP32
void ref_get_cross_kv( whisper_context* ctx, int layer, float* k, float* v )
{
	const auto& h = ctx->model.hparams;
	const size_t n = (size_t)h.n_audio_ctx * h.n_audio_state;
	copyF16( ctx->model.memory_cross_k, n * layer, n, k );
	copyF16( ctx->model.memory_cross_v, n * layer, n, v );
}
// self-attention cache rows [0, nRows) of one decoder layer, [row][n_state] (whisper.cpp:1609-1613)
void ref_get_self_kv( whisper_context* ctx, int layer, int nRows, float* k, float* v )
{
	const auto& h = ctx->model.hparams;
	const size_t off = (size_t)h.n_text_ctx * h.n_text_state * layer;
	copyF16( ctx->model.memory_k, off, (size_t)nRows * h.n_text_state, k );
	copyF16( ctx->model.memory_v, off, (size_t)nRows * h.n_text_state, v );
}

// ---- probe-point capture ----
void ref_trace_enable( int on ) { Tracing::g_enabled = on != 0; if( !on ) Tracing::g_captured.clear(); }
void ref_trace_clear() { Tracing::g_captured.clear(); }
int ref_trace_count() { return (int)Tracing::g_captured.size(); }
int ref_trace_name( int idx, char* dst, int cap )
{
	int i = 0;
	for( const auto& kv : Tracing::g_captured )
		if( i++ == idx ) { snprintf( dst, cap, "%s", kv.first.c_str() ); return 0; }
	return -1;
}
// returns element count, fills ne[4]; dst may be null to query
long ref_trace_get( const char* name, int32_t* ne4, float* dst, long cap )
{
	auto it = Tracing::g_captured.find( name );
	if( it == Tracing::g_captured.end() ) return -1;
	if( ne4 ) memcpy( ne4, it->second.ne, 16 );
	const long n = (long)it->second.data.size();
	if( dst && cap >= n ) memcpy( dst, it->second.data.data(), n * sizeof( float ) );
	return n;
}

// ---- sampling / vocabulary helpers (whisper.cpp:1875-1960, 2513-2556) ----
void ref_sample_best( whisper_context* ctx, int32_t* id, int32_t* tid, float* p, float* pt, float* ptsum )
{
	const whisper_token_data d = whisper_sample_best( ctx );
	*id = d.id; *tid = d.tid; *p = d.p; *pt = d.pt; *ptsum = d.ptsum;
}
void ref_sample_timestamp( whisper_context* ctx, int isInitial, int32_t* id, int32_t* tid, float* p, float* pt, float* ptsum )
{
	const whisper_token_data d = whisper_sample_timestamp( ctx, isInitial != 0 );
	*id = d.id; *tid = d.tid; *p = d.p; *pt = d.pt; *ptsum = d.ptsum;
}
int ref_token_special( whisper_context* ctx, int which )
{
	switch( which )
	{
	case 0: return whisper_token_eot( ctx );
	case 1: return whisper_token_sot( ctx );
	case 2: return whisper_token_prev( ctx );
	case 3: return whisper_token_solm( ctx );
	case 4: return whisper_token_not( ctx );
	case 5: return whisper_token_beg( ctx );
	case 6: return whisper_token_translate();
	case 7: return whisper_token_transcribe();
	}
	return -1;
}
const char* ref_token_to_str( whisper_context* ctx, int token ) { return whisper_token_to_str( ctx, token ); }
int ref_tokenize( whisper_context* ctx, const char* text, int32_t* tokens, int cap ) { return whisper_tokenize( ctx, text, tokens, cap ); }
int ref_is_multilingual( whisper_context* ctx ) { return whisper_is_multilingual( ctx ); }

// ---- whisper_full, greedy (whisper.cpp:2765-3120): the reference's complete runFull equivalent ----
// flags: bit0 = no_context, bit1 = single_segment, bit2 = translate
int ref_full( whisper_context* ctx, const float* pcm, int nSamples, int nThreads, const char* lang, int flags, int maxTokens,
	int audioCtx, const int32_t* promptTokens, int nPrompt, int nMaxTextCtx )
{
	whisper_full_params p = whisper_full_default_params( WHISPER_SAMPLING_GREEDY );
	p.n_threads = nThreads;
	p.print_progress = false;
	p.print_realtime = false;
	p.print_timestamps = false;
	p.print_special = false;
	p.language = lang;
	p.no_context = ( flags & 1 ) != 0;
	p.single_segment = ( flags & 2 ) != 0;
	p.translate = ( flags & 4 ) != 0;
	p.max_tokens = maxTokens;
	p.audio_ctx = audioCtx;
	p.prompt_tokens = promptTokens;
	p.prompt_n_tokens = nPrompt;
	if( nMaxTextCtx >= 0 ) p.n_max_text_ctx = nMaxTextCtx;
	return whisper_full( ctx, p, pcm, nSamples );
}
// the same with a range of the audio (offset_ms / duration_ms, whisper.cpp:2786-2787) and print_special
int ref_full_range( whisper_context* ctx, const float* pcm, int nSamples, int nThreads, const char* lang, int flags, int maxTokens,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx, int offsetMs, int durationMs )
{
	whisper_full_params p = whisper_full_default_params( WHISPER_SAMPLING_GREEDY );
	p.n_threads = nThreads;
	p.print_progress = false;
	p.print_realtime = false;
	p.print_timestamps = false;
	p.print_special = ( flags & 8 ) != 0;
	p.language = lang;
	p.no_context = ( flags & 1 ) != 0;
	p.single_segment = ( flags & 2 ) != 0;
	p.translate = ( flags & 4 ) != 0;
	p.max_tokens = maxTokens;
	p.prompt_tokens = promptTokens;
	p.prompt_n_tokens = nPrompt;
	if( nMaxTextCtx >= 0 ) p.n_max_text_ctx = nMaxTextCtx;
	p.offset_ms = offsetMs;
	p.duration_ms = durationMs;
	return whisper_full( ctx, p, pcm, nSamples );
}
// whisper_full with token-level timestamps (whisper.cpp:2803-2808, 3063-3069): thold_pt / thold_ptsum / max_len as given
int ref_full_token_timestamps( whisper_context* ctx, const float* pcm, int nSamples, int nThreads, const char* lang, int flags,
	const int32_t* promptTokens, int nPrompt, int nMaxTextCtx, float tholdPt, float tholdPtsum, int maxLen )
{
	whisper_full_params p = whisper_full_default_params( WHISPER_SAMPLING_GREEDY );
	p.n_threads = nThreads;
	p.print_progress = false;
	p.print_realtime = false;
	p.print_timestamps = false;
	p.print_special = false;
	p.language = lang;
	p.no_context = ( flags & 1 ) != 0;
	p.single_segment = ( flags & 2 ) != 0;
	p.translate = ( flags & 4 ) != 0;
	p.prompt_tokens = promptTokens;
	p.prompt_n_tokens = nPrompt;
	if( nMaxTextCtx >= 0 ) p.n_max_text_ctx = nMaxTextCtx;
	p.token_timestamps = true;
	p.thold_pt = tholdPt;
	p.thold_ptsum = tholdPtsum;
	p.max_len = maxLen;
	return whisper_full( ctx, p, pcm, nSamples );
}
// t[2] = { t0, t1 } in 10 ms units, f[4] = { p, pt, ptsum, vlen }
void ref_full_token_data( whisper_context* ctx, int i, int j, int64_t* t, float* f )
{
	const whisper_token_data d = whisper_full_get_token_data( ctx, i, j );
	t[ 0 ] = d.t0; t[ 1 ] = d.t1;
	f[ 0 ] = d.p; f[ 1 ] = d.pt; f[ 2 ] = d.ptsum; f[ 3 ] = d.vlen;
}
int ref_full_n_segments( whisper_context* ctx ) { return whisper_full_n_segments( ctx ); }
int64_t ref_full_segment_t0( whisper_context* ctx, int i ) { return whisper_full_get_segment_t0( ctx, i ); }
int64_t ref_full_segment_t1( whisper_context* ctx, int i ) { return whisper_full_get_segment_t1( ctx, i ); }
const char* ref_full_segment_text( whisper_context* ctx, int i ) { return whisper_full_get_segment_text( ctx, i ); }
int ref_full_n_tokens( whisper_context* ctx, int i ) { return whisper_full_n_tokens( ctx, i ); }
int ref_full_token_id( whisper_context* ctx, int i, int j ) { return whisper_full_get_token_id( ctx, i, j ); }
float ref_full_token_p( whisper_context* ctx, int i, int j ) { return whisper_full_get_token_p( ctx, i, j ); }
int ref_full_token_tid( whisper_context* ctx, int i, int j ) { return whisper_full_get_token_data( ctx, i, j ).tid; }

// timing counters the reference keeps itself (whisper.cpp:2557-2568), microseconds
void ref_timings( whisper_context* ctx, int64_t* out5 )
{
	out5[ 0 ] = ctx->t_load_us; out5[ 1 ] = ctx->t_mel_us; out5[ 2 ] = ctx->t_sample_us;
	out5[ 3 ] = ctx->t_encode_us; out5[ 4 ] = ctx->t_decode_us;
}
void ref_reset_timings( whisper_context* ctx ) { whisper_reset_timings( ctx ); }
const char* ref_system_info() { return whisper_print_system_info(); }

// the two 65536-entry FP16 lookup tables built by ggml_init (ggml.c:1375-1385); exported as uint16 bit patterns
void ref_lookup_tables( uint16_t* gelu, uint16_t* expo )
{
	struct ggml_init_params ip;
	ip.mem_size = 1 << 20;
	ip.mem_buffer = nullptr;
	ggml_context* c = ggml_init( ip );
	for( int i = 0; i < 65536; i++ )
	{
		// probe through the public conversion API: table[i] == fp16( f( fp32(i) ) ), see ggml.c:1381-1382
		ggml_fp16_t h; uint16_t u = (uint16_t)i; memcpy( &h, &u, 2 );
		const float f = ggml_fp16_to_fp32( h );
		ggml_fp16_t g = ggml_fp32_to_fp16( (float)( 0.5 * f * ( 1.0 + tanh( 0.79788456080286535587989211986876 * f * ( 1.0 + 0.044715 * f * f ) ) ) ) );
		ggml_fp16_t e = ggml_fp32_to_fp16( (float)exp( f ) );
		memcpy( gelu + i, &g, 2 ); memcpy( expo + i, &e, 2 );
	}
	ggml_free( c );
}

}	// extern "C"
