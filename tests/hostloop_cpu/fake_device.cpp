// tests/hostloop_cpu/fake_device.cpp -- a TEST DOUBLE of the compute layer, for CPU tests of host code only. Never shipped, never loaded by the
// product: it is compiled INTO the test library tests/_build/libbatch_cpu.so next to whisper_amd/host/batchScheduler.cpp (no libwhisper_hip.so,
// no libWhisper.so involved), so that the lock-step scheduler's logic -- slots, refills, groups, ragged prompts, chunked fetches, retirement,
// failure paths -- runs without a GPU. It implements the dozen entry points of include/whisper_hip.h that scheduler calls, with the reference's
// own CPU model (oracle/_ref/libwhisper_ref.so: whisper_encode / whisper_decode / whisper_sample_best) as the arithmetic: one whisper_context
// per slot plays that slot's rows of the device batch. "Device" buffers are host memory, every call completes before it returns.
// The entry points the scheduler does not call but support.cpp's model loader references are present and fail.
#include "whisper_hip.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
void* ref_init( const char* path );
void ref_free( void* ctx );
void ref_set_log_level( int lvl );
int ref_pcm_to_mel( void* ctx, const float* pcm, int n, int nThreads );
int ref_mel_len( void* ctx );
void ref_get_mel( void* ctx, float* dst );
int ref_set_mel( void* ctx, const float* mel, int nLen, int nMel );
int ref_encode( void* ctx, int melOffset, int nThreads );
int ref_decode( void* ctx, const int32_t* tokens, int nTokens, int nPast, int nThreads );
void ref_sample_best( void* ctx, int32_t* id, int32_t* tid, float* p, float* pt, float* ptsum );
void ref_sample_timestamp( void* ctx, int isInitial, int32_t* id, int32_t* tid, float* p, float* pt, float* ptsum );
}

struct wh_model
{
	std::string path;	 // the ggml file every slot's CPU model is loaded from
	int threads = 4;
};
struct wh_context
{
	wh_model* model = nullptr;
	std::vector<void*> cpu;			  // one whisper_context per slot
	std::vector<char> active;		  // slot has a window (melDev != NULL in the last wh_encode_windows)
	std::vector<int> nPast, last;
	std::vector<std::vector<wh_token_data>> samples;	// per slot, the window in progress
	int batch = 0;
	// counters a test can read: how the scheduler used the device
	int encodes = 0, windows = 0, starts = 0, continues = 0, stepsDecoded = 0;
};

namespace
{
	std::vector<wh_context*> g_contexts;
	void* slot( wh_context* c, int b )
	{
		if( !c->cpu[ b ] ) c->cpu[ b ] = ref_init( c->model->path.c_str() );
		return c->cpu[ b ];
	}
	void sample( wh_context* c, int b, bool initialTimestamp )
	{
		wh_token_data t{};
		if( initialTimestamp ) ref_sample_timestamp( c->cpu[ b ], 1, &t.id, &t.tid, &t.p, &t.pt, &t.ptsum );
		else ref_sample_best( c->cpu[ b ], &t.id, &t.tid, &t.p, &t.pt, &t.ptsum );
		c->samples[ b ].push_back( t );
		c->last[ b ] = t.id;
	}
	int steps( wh_context* c, int n )
	{
		for( int b = 0; b < c->batch; b++ )
		{
			if( !c->active[ b ] )
			{
				c->samples[ b ].resize( c->samples[ b ].size() + (size_t)n, wh_token_data{} );
				continue;
			}
			for( int s = 0; s < n; s++ )
			{
				const int32_t tok = c->last[ b ];
				if( 0 != ref_decode( c->cpu[ b ], &tok, 1, c->nPast[ b ], c->model->threads ) ) return -1;
				c->nPast[ b ] += 1;
				sample( c, b, false );
				c->stepsDecoded++;
			}
		}
		return 0;
	}
}

extern "C" {
// ---- test hooks (not part of include/whisper_hip.h) ----
wh_model* fake_model_create( const char* ggmlPath, int threads )
{
	ref_set_log_level( 0 );
	wh_model* m = new wh_model();
	m->path = ggmlPath;
	m->threads = threads;
	return m;
}
// sums over the contexts created so far: { contexts, encoder batches, windows encoded, window starts, continues, decode steps }
void fake_device_counters( int64_t* out6 )
{
	int64_t v[ 6 ] = { (int64_t)g_contexts.size(), 0, 0, 0, 0, 0 };
	for( wh_context* c : g_contexts ) { v[ 1 ] += c->encodes; v[ 2 ] += c->windows; v[ 3 ] += c->starts; v[ 4 ] += c->continues; v[ 5 ] += c->stepsDecoded; }
	memcpy( out6, v, sizeof( v ) );
}

const char* wh_last_error( void ) { return "fake device"; }
void wh_model_destroy( wh_model* m ) { delete m; }

// the double decodes with the reference CPU model at the model's own n_audio_ctx: an override is refused (the device-side override has its own GPU test)
int wh_context_set_audio_ctx( wh_context* c, int audioCtx ) { return !c ? WH_E_INVALIDARG : ( audioCtx == 0 ? 0 : WH_E_INVALIDARG ); }

int wh_context_create( wh_model* m, int maxBatch, void*, wh_context** out )
{
	if( !m || maxBatch <= 0 || !out ) return WH_E_INVALIDARG;
	wh_context* c = new wh_context();
	c->model = m;
	c->cpu.assign( (size_t)maxBatch, nullptr );
	c->active.assign( (size_t)maxBatch, 0 );
	c->nPast.assign( (size_t)maxBatch, 0 );
	c->last.assign( (size_t)maxBatch, 0 );
	c->samples.resize( (size_t)maxBatch );
	g_contexts.push_back( c );
	*out = c;
	return 0;
}
void wh_context_destroy( wh_context* c )
{
	if( !c ) return;
	for( void* p : c->cpu )
		if( p ) ref_free( p );
	for( size_t i = 0; i < g_contexts.size(); i++ )
		if( g_contexts[ i ] == c ) { g_contexts.erase( g_contexts.begin() + i ); break; }
	delete c;
}
int wh_context_bind( wh_context* ) { return 0; }
int wh_context_synchronize( wh_context* ) { return 0; }
int wh_buffer_alloc( int64_t bytes, void** dev ) { *dev = malloc( (size_t)( bytes > 0 ? bytes : 1 ) ); return *dev ? 0 : WH_E_OUTOFMEMORY; }
int wh_buffer_free( void* dev ) { free( dev ); return 0; }
int wh_buffer_upload_async( wh_context*, void* dev, const void* host, int64_t bytes ) { memcpy( dev, host, (size_t)bytes ); return 0; }

// the whole-buffer spectrogram of a stream (Spectrogram::pcmToMel == log_mel_spectrogram): slot 0's CPU model computes it
int wh_mel_spectrogram( wh_context* c, const float* pcmDev, int64_t nSamples, float* melDev, int64_t* nLenOut )
{
	void* w = slot( c, 0 );
	if( 0 != ref_pcm_to_mel( w, pcmDev, (int)nSamples, c->model->threads ) ) return -1;
	ref_get_mel( w, melDev );
	if( nLenOut ) *nLenOut = ref_mel_len( w );
	return 0;
}
int wh_encode_windows( wh_context* c, const wh_mel_window* windows, int batch )
{
	if( batch > (int)c->cpu.size() ) return WH_E_INVALIDARG;
	c->encodes++;
	c->batch = batch;
	for( int b = 0; b < batch; b++ )
	{
		c->active[ b ] = windows[ b ].melDev != nullptr;
		if( !c->active[ b ] ) continue;
		void* w = slot( c, b );
		if( 0 != ref_set_mel( w, windows[ b ].melDev, (int)windows[ b ].melLen, 80 ) ) return -1;
		if( 0 != ref_encode( w, windows[ b ].offset, c->model->threads ) ) return -1;
		c->windows++;
	}
	return 0;
}
int wh_decode_window_start_ragged( wh_context* c, int batch, const int32_t* promptTokens, const int32_t* promptLens, int nPromptMax, int nSteps, int, int )
{
	if( batch != c->batch ) return WH_E_INVALIDARG;
	c->starts++;
	for( int b = 0; b < batch; b++ )
	{
		c->samples[ b ].clear();
		if( !c->active[ b ] ) { c->samples[ b ].push_back( wh_token_data{} ); continue; }
		if( promptLens[ b ] < 1 || promptLens[ b ] > nPromptMax ) return WH_E_INVALIDARG;
		if( 0 != ref_decode( c->cpu[ b ], promptTokens + (size_t)b * nPromptMax, promptLens[ b ], 0, c->model->threads ) ) return -1;
		c->nPast[ b ] = promptLens[ b ];
		sample( c, b, true );
	}
	return steps( c, nSteps );
}
int wh_decode_window_continue( wh_context* c, int nSteps ) { c->continues++; return steps( c, nSteps ); }
int wh_decode_window_ready( wh_context* c, int first, int count ) { return ( c->batch > 0 && (int)c->samples[ 0 ].size() >= first + count ) ? 1 : 0; }
int wh_decode_window_fetch( wh_context* c, int first, int count, wh_token_data* out )
{
	for( int b = 0; b < c->batch; b++ )
	{
		if( (int)c->samples[ b ].size() < first + count ) return WH_E_INVALIDARG;
		for( int k = 0; k < count; k++ ) out[ (size_t)k * c->batch + b ] = c->samples[ b ][ (size_t)first + k ];
	}
	return 0;
}

// ---- referenced by support.cpp's model loader (loadGgmlFile, listGPUs), which these tests never call ----
int wh_device_set( int ) { return WH_E_NO_DEVICE; }
int wh_device_count( void ) { return 0; }
int wh_device_info( int, char*, size_t, uint64_t*, int* ) { return WH_E_NO_DEVICE; }
int wh_model_create( const wh_hparams*, void*, int, wh_model** ) { return WH_E_NO_DEVICE; }
int wh_model_set_filters( wh_model*, int, int, const float* ) { return WH_E_NO_DEVICE; }
int wh_model_set_tensor( wh_model*, const char*, int, const int32_t*, int, const void* ) { return WH_E_NO_DEVICE; }
int wh_model_finalize( wh_model* ) { return WH_E_NO_DEVICE; }
int wh_model_broadcast( wh_model*, wh_comm*, int, double* ) { return WH_E_NO_DEVICE; }
int wh_model_arena( wh_model*, void**, int64_t* ) { return WH_E_NO_DEVICE; }
int wh_comm_info( const wh_comm*, int*, int* ) { return WH_E_NO_DEVICE; }
int wh_comm_broadcast_i32( wh_comm*, int, int32_t* ) { return WH_E_NO_DEVICE; }
}
