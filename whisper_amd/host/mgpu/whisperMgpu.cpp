// One process per GPU over the C ABI, no Python and no torch: the multi-GPU harness of the drop-in library.
//
//   whisper-mgpu -n 8 -m ggml-medium.bin -f recording.wav [-l en] [-o out.txt]
//
// The parent forks N ranks (or, when RANK / WORLD_SIZE / LOCAL_RANK are set by an external launcher, runs as that rank).
// Rank r binds GPU LOCAL_RANK % devices (sModelSetup.adapter); rank 0 creates the RCCL id and publishes it through a file
// (-id, default a temp file); every rank builds its communicator, the model is read by rank 0 only and broadcast over xGMI
// (loadModelShared, Whisper/Whisper/ModelImpl.cpp:40-60 is the reference's single-GPU counterpart), and the recording's
// 30 s windows are split into contiguous ranges, one per rank (eFullParamsFlags::NoContext: windows are independent, the
// only exchange is the broadcast). Each rank writes "<out>.rank<r>"; the parent concatenates them in rank order.
// The same sharding as whisper_amd/distributed.py (shard_range) and bench.py --gpus N.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include "whisperApi.h"
#include "whisper_hip.h"
using namespace Whisper;

namespace
{
	struct Args
	{
		int ranks = 1;
		std::string model, wav, lang = "en", out = "transcript.txt", idFile;
	};
	std::wstring widen( const std::string& s )
	{
		std::wstring w;
		for( unsigned char c : s ) w += (wchar_t)c;
		return w;
	}
	// contiguous balanced ranges, the first (count % world) ranks take one more: whisper_amd/distributed.py shard_range
	void shardRange( int count, int rank, int world, int& b, int& e )
	{
		const int q = count / world, r = count % world;
		b = rank * q + ( rank < r ? rank : r );
		e = b + q + ( rank < r ? 1 : 0 );
	}
	bool readFile( const std::string& path, void* dst, size_t n )
	{
		FILE* f = fopen( path.c_str(), "rb" );
		if( !f ) return false;
		const size_t got = fread( dst, 1, n, f );
		fclose( f );
		return got == n;
	}

	int runRank( const Args& a, int rank, int world, int localRank )
	{
		const auto t0 = std::chrono::steady_clock::now();
		auto since = [ & ]() { return std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count(); };
		const int nDev = wh_device_count();
		if( nDev <= 0 ) { fprintf( stderr, "[rank %d] no GPU\n", rank ); return 2; }
		const int device = localRank % nDev;
		if( 0 != wh_device_set( device ) ) { fprintf( stderr, "[rank %d] %s\n", rank, wh_last_error() ); return 2; }

		// ---- communicator: the 128-byte id travels through a file ----
		unsigned char id[ WH_COMM_ID_BYTES ];
		const std::string tmp = a.idFile + ".tmp";
		if( rank == 0 )
		{
			if( 0 != wh_comm_unique_id( id ) ) { fprintf( stderr, "[rank 0] %s\n", wh_last_error() ); return 3; }
			FILE* f = fopen( tmp.c_str(), "wb" );
			if( !f || fwrite( id, 1, sizeof( id ), f ) != sizeof( id ) ) { fprintf( stderr, "[rank 0] cannot write %s\n", tmp.c_str() ); return 3; }
			fclose( f );
			rename( tmp.c_str(), a.idFile.c_str() );	// atomic publish
		}
		else
		{
			int waited = 0;
			while( !readFile( a.idFile, id, sizeof( id ) ) )
			{
				if( ++waited > 6000 ) { fprintf( stderr, "[rank %d] no communicator id after 60 s\n", rank ); return 3; }
				std::this_thread::sleep_for( std::chrono::milliseconds( 10 ) );
			}
		}
		wh_comm* comm = nullptr;
		if( 0 != wh_comm_create( id, rank, world, &comm ) ) { fprintf( stderr, "[rank %d] %s\n", rank, wh_last_error() ); return 3; }
		fprintf( stderr, "[rank %d/%d] device %d, communicator up after %.2f s\n", rank, world, device, since() );

		// ---- model: rank 0 reads, everyone receives ----
		const std::wstring adapter = std::to_wstring( device ) + L":";
		sModelSetup setup;
		setup.adapter = adapter.c_str();
		iModel* model = nullptr;
		HRESULT hr = loadModelShared( widen( a.model ).c_str(), setup, nullptr, comm, 0, &model );
		if( FAILED( hr ) ) { fprintf( stderr, "[rank %d] loadModelShared failed 0x%08x\n", rank, (unsigned)hr ); return 4; }
		fprintf( stderr, "[rank %d] model ready after %.2f s\n", rank, since() );

		// ---- audio: every rank decodes the file (16 kHz PCM is small next to the model), then takes its windows ----
		iMediaFoundation* mf = nullptr;
		iAudioBuffer* audio = nullptr;
		hr = initMediaFoundation( &mf );
		if( SUCCEEDED( hr ) ) hr = mf->loadAudioFile( a.wav.c_str(), false, &audio );
		if( FAILED( hr ) ) { fprintf( stderr, "[rank %d] cannot load %s (0x%08x)\n", rank, a.wav.c_str(), (unsigned)hr ); return 5; }
		const uint32_t nSamples = audio->countSamples();
		const int windows = (int)( ( nSamples + 16000 * 30 - 1 ) / ( 16000 * 30 ) );
		int wb = 0, we = 0;
		shardRange( windows, rank, world, wb, we );

		iContext* ctx = nullptr;
		hr = model->createContext( &ctx );
		if( FAILED( hr ) ) return 6;
		sFullParams p;
		ctx->fullDefaultParams( eSamplingStrategy::Greedy, &p );
		p.language = findLanguageKeyA( a.lang.c_str() );
		p.setFlag( eFullParamsFlags::NoContext );
		p.setFlag( eFullParamsFlags::PrintRealtime, false );
		p.setFlag( eFullParamsFlags::PrintProgress, false );
		p.offset_ms = wb * 30000;
		p.duration_ms = we == windows ? 0 : ( we - wb ) * 30000;	   // 0 = to the end of the recording (the last range is ragged)
		wh_comm_barrier( comm );
		const double tStart = since();
		std::string text;
		int nSeg = 0;
		if( we > wb )
		{
			hr = ctx->runFull( p, audio );
			if( FAILED( hr ) ) { fprintf( stderr, "[rank %d] runFull failed 0x%08x\n", rank, (unsigned)hr ); return 7; }
			iTranscribeResult* res = nullptr;
			if( SUCCEEDED( ctx->getResults( eResultFlags::Timestamps, &res ) ) && res )
			{
				sTranscribeLength len;
				res->getSize( len );
				const sSegment* seg = res->getSegments();
				for( uint32_t i = 0; i < len.countSegments; i++ )
				{
					char line[ 96 ];
					snprintf( line, sizeof( line ), "[%9.2f --> %9.2f] ", seg[ i ].time.begin.ticks / 1e7, seg[ i ].time.end.ticks / 1e7 );
					text += line;
					text += seg[ i ].text ? seg[ i ].text : "";
					text += "\n";
				}
				nSeg = (int)len.countSegments;
				res->Release();
			}
		}
		const double tRun = since() - tStart;
		wh_comm_barrier( comm );
		const double tAll = since() - tStart;
		FILE* f = fopen( ( a.out + ".rank" + std::to_string( rank ) ).c_str(), "wb" );
		if( f ) { fwrite( text.data(), 1, text.size(), f ); fclose( f ); }
		fprintf( stderr, "[rank %d] windows %d..%d of %d: %d segments in %.3f s (%.1f audio-s/s on this rank); all ranks done after %.3f s\n", rank, wb, we, windows,
			nSeg, tRun, tRun > 0 ? ( we - wb ) * 30.0 / tRun : 0.0, tAll );
		if( rank == 0 )
		{
			printf( "{\"ranks\": %d, \"windows\": %d, \"seconds\": %.4f, \"audio_seconds_per_sec\": %.2f}\n", world, windows, tAll, windows * 30.0 / tAll );
			fflush( stdout );	   // the rank leaves through _exit
		}
		ctx->Release();
		audio->Release();
		mf->Release();
		model->Release();
		wh_comm_destroy( comm );
		return 0;
	}
}	// namespace

int main( int argc, char** argv )
{
	Args a;
	for( int i = 1; i < argc; i++ )
	{
		auto val = [ & ]() -> const char* { return i + 1 < argc ? argv[ ++i ] : ""; };
		if( !strcmp( argv[ i ], "-n" ) ) a.ranks = atoi( val() );
		else if( !strcmp( argv[ i ], "-m" ) ) a.model = val();
		else if( !strcmp( argv[ i ], "-f" ) ) a.wav = val();
		else if( !strcmp( argv[ i ], "-l" ) ) a.lang = val();
		else if( !strcmp( argv[ i ], "-o" ) ) a.out = val();
		else if( !strcmp( argv[ i ], "-id" ) ) a.idFile = val();
		else if( !strcmp( argv[ i ], "--shard-range" ) )
		{
			// test hook (no GPU): the window range rank r of w gets out of n windows, "begin end" -- must equal whisper_amd/distributed.py shard_range
			const int n = atoi( val() ), r = atoi( val() ), w = atoi( val() );
			if( n < 0 || w < 1 || r < 0 || r >= w ) return 1;
			int b = 0, e = 0;
			shardRange( n, r, w, b, e );
			printf( "%d %d\n", b, e );
			return 0;
		}
		else { fprintf( stderr, "usage: whisper-mgpu -n ranks -m model.bin -f audio.wav [-l en] [-o out.txt] [-id id-file]\n" ); return 1; }
	}
	if( a.model.empty() || a.wav.empty() || a.ranks < 1 ) { fprintf( stderr, "whisper-mgpu: -m and -f are required\n" ); return 1; }
	if( a.idFile.empty() ) a.idFile = "/tmp/whisper-mgpu." + std::to_string( (long)getpid() ) + ".id";

	// launched by torchrun / mpirun / srun: be the rank the environment names
	if( const char* wr = getenv( "RANK" ) )
	{
		const int world = getenv( "WORLD_SIZE" ) ? atoi( getenv( "WORLD_SIZE" ) ) : 1;
		const int local = getenv( "LOCAL_RANK" ) ? atoi( getenv( "LOCAL_RANK" ) ) : atoi( wr );
		return runRank( a, atoi( wr ), world, local );
	}
	unlink( a.idFile.c_str() );
	std::vector<pid_t> kids;
	for( int r = 0; r < a.ranks; r++ )
	{
		const pid_t pid = fork();	// before any HIP call: a forked HIP runtime is not usable
		if( pid == 0 ) _exit( runRank( a, r, a.ranks, r ) );
		if( pid < 0 ) { perror( "fork" ); return 1; }
		kids.push_back( pid );
	}
	int rc = 0;
	for( pid_t k : kids )
	{
		int st = 0;
		waitpid( k, &st, 0 );
		if( !WIFEXITED( st ) || WEXITSTATUS( st ) != 0 ) rc = 1;
	}
	unlink( a.idFile.c_str() );
	if( rc == 0 )
	{
		FILE* out = fopen( a.out.c_str(), "wb" );
		for( int r = 0; out && r < a.ranks; r++ )
		{
			const std::string part = a.out + ".rank" + std::to_string( r );
			if( FILE* f = fopen( part.c_str(), "rb" ) )
			{
				char buf[ 65536 ];
				size_t n;
				while( ( n = fread( buf, 1, sizeof( buf ), f ) ) > 0 ) fwrite( buf, 1, n, out );
				fclose( f );
				unlink( part.c_str() );
			}
		}
		if( out ) fclose( out );
	}
	return rc;
}
