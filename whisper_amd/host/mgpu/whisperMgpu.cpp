// One process per GPU over the C ABI, no Python and no torch: the multi-GPU harness of the drop-in library.
//
//   whisper-mgpu -n 8 -m ggml-medium.bin -f recording.wav [-l en] [-o out.txt] [-timeout 300] [-job-timeout 0] [-slots 64] [-id file] [-job token]
//   whisper-mgpu -n 8 -m ggml-medium.bin -per-recording -f a.wav -f b.wav ...   whole recordings, one stream each with the reference's sliding window (no words cut at 30 s)
//
// The parent forks N ranks (or, when RANK / WORLD_SIZE / LOCAL_RANK are set by an external launcher, runs as that rank).
// Rank r binds GPU LOCAL_RANK % devices (sModelSetup.adapter); rank 0 creates the RCCL id and publishes it through a file;
// every rank builds its communicator, the model is read by rank 0 only and broadcast over xGMI (loadModelShared;
// Whisper/Whisper/ModelImpl.cpp:40-60 is the reference's single-GPU counterpart), and the recording is cut into independent 30 s
// chunks (north_star's sharding unit) dealt to the ranks as contiguous ranges -- the only exchange is the broadcast.
//
// A chunk is a recording of its own (sBatchStream::firstSample / countSamples: own spectrogram maximum, nothing read past its end,
// times shifted by its start), and a rank transcribes ITS chunks as one Whisper::runFullBatch call -- lock-step batches on its GPU.
// So the concatenated transcript does not depend on the number of ranks: N = 8 prints what N = 1 prints. (Round 3 gave each rank one
// sequential runFull over offset_ms / duration_ms: a rank's last window read past its range and the next rank started mid-utterance.)
//
// No rank may hang the node: the rendezvous and every collective carry a deadline (wh_comm_create_timeout / wh_comm_set_timeout), a
// root that cannot read the model says so to the ranks about to enter the broadcast (loadModelShared), and the parent reaps whichever
// child ends first -- a non-zero exit or the job's deadline ends the remaining ranks (SIGTERM, then SIGKILL).
// The same dealing of chunks as whisper_amd/distributed.py (shard_range) and bench.py --gpus N.
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
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
		int ranks = 1, slots = 64;
		double timeout = 300.0;		// deadline of the rendezvous and of every collective
		double jobTimeout = 0.0;	// deadline of the whole job (forked mode: the parent ends the ranks); 0 = none
		std::string model, wav, lang = "en", out = "transcript.txt", idFile;
		std::vector<std::string> wavs;	// every -f
		bool perRecording = false;	// -per-recording: whole recordings dealt round-robin to the ranks, each decoded with the reference's sliding window
		std::string job;			// token of THIS job: stamped into the id file by rank 0, required by the ranks that read it
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
	// The id file: a header naming the job, then the 128-byte id. A rank takes the file only when it carries ITS job's token -- not when it merely
	// looks recent: under an external launcher (torchrun / srun / mpirun) a rank may start many seconds after rank 0 published the id, so
	// freshness is judged against the rendezvous deadline (a file older than `-timeout` before this rank started cannot be this job's: rank 0
	// would have given up by now), never against this rank's own start time (that rejected the valid file on every poll).
	struct IdFileHeader
	{
		char magic[ 8 ];	 // "WHMGPU1"
		char job[ 56 ];		 // NUL-padded token
	};
	IdFileHeader makeHeader( const std::string& job )
	{
		IdFileHeader h;
		memset( &h, 0, sizeof( h ) );
		memcpy( h.magic, "WHMGPU1", 7 );
		strncpy( h.job, job.c_str(), sizeof( h.job ) - 1 );
		return h;
	}
	bool writeIdFile( const std::string& path, const std::string& job, const void* id, size_t n )
	{
		const std::string tmp = path + ".tmp";
		FILE* f = fopen( tmp.c_str(), "wb" );
		if( !f ) return false;
		const IdFileHeader h = makeHeader( job );
		const bool ok = fwrite( &h, 1, sizeof( h ), f ) == sizeof( h ) && fwrite( id, 1, n, f ) == n;
		fclose( f );
		if( !ok ) { unlink( tmp.c_str() ); return false; }
		return 0 == rename( tmp.c_str(), path.c_str() );	// atomic publish
	}
	bool readIdFile( const std::string& path, const std::string& job, void* dst, size_t n, time_t notBefore )
	{
		struct stat st;
		if( 0 != stat( path.c_str(), &st ) || st.st_mtime < notBefore ) return false;
		FILE* f = fopen( path.c_str(), "rb" );
		if( !f ) return false;
		IdFileHeader h;
		const bool ok = fread( &h, 1, sizeof( h ), f ) == sizeof( h ) && fread( dst, 1, n, f ) == n;
		fclose( f );
		const IdFileHeader want = makeHeader( job );
		return ok && 0 == memcmp( &h, &want, sizeof( h ) );
	}
	// WHISPER_MGPU_TEST_FAULT="<rank|all>:<exit|hang>" -- test hook (tests/test_cli.py, no GPU needed): the named rank exits with code 7 or
	// sleeps for ever before it touches a device, so that the parent's handling of a dead / stuck rank can be exercised anywhere
	void injectedFault( int rank )
	{
		const char* e = getenv( "WHISPER_MGPU_TEST_FAULT" );
		if( !e ) return;
		const char* colon = strchr( e, ':' );
		if( !colon ) return;
		const bool mine = 0 == strncmp( e, "all", 3 ) || atoi( e ) == rank;
		if( !mine ) return;
		if( 0 == strcmp( colon + 1, "exit" ) ) _exit( 7 );
		if( 0 == strcmp( colon + 1, "hang" ) )
			while( true ) sleep( 1000 );
	}

	int runRank( const Args& a, int rank, int world, int localRank )
	{
		injectedFault( rank );
		const time_t started = time( nullptr );
		const auto t0 = std::chrono::steady_clock::now();
		auto since = [ & ]() { return std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count(); };
		const int nDev = wh_device_count();
		if( nDev <= 0 ) { fprintf( stderr, "[rank %d] no GPU\n", rank ); return 2; }
		const int device = localRank % nDev;
		if( 0 != wh_device_set( device ) ) { fprintf( stderr, "[rank %d] %s\n", rank, wh_last_error() ); return 2; }

		// ---- communicator: the 128-byte id travels through a file ----
		unsigned char id[ WH_COMM_ID_BYTES ];
		if( rank == 0 )
		{
			unlink( a.idFile.c_str() );	   // whatever an earlier job left under this name
			if( 0 != wh_comm_unique_id( id ) ) { fprintf( stderr, "[rank 0] %s\n", wh_last_error() ); return 3; }
			if( !writeIdFile( a.idFile, a.job, id, sizeof( id ) ) ) { fprintf( stderr, "[rank 0] cannot write %s\n", a.idFile.c_str() ); return 3; }
		}
		else
		{
			// this job's token, and not older than the rendezvous deadline allows (see readIdFile)
			while( !readIdFile( a.idFile, a.job, id, sizeof( id ), started - (time_t)a.timeout - 2 ) )
			{
				if( since() > a.timeout ) { fprintf( stderr, "[rank %d] no communicator id of job '%s' in %s after %.0f s\n", rank, a.job.c_str(), a.idFile.c_str(), a.timeout ); return 3; }
				std::this_thread::sleep_for( std::chrono::milliseconds( 10 ) );
			}
		}
		wh_comm* comm = nullptr;
		if( 0 != wh_comm_create_timeout( id, rank, world, a.timeout, &comm ) ) { fprintf( stderr, "[rank %d] %s\n", rank, wh_last_error() ); return 3; }
		int seenRank = -1, seenWorld = -1;
		wh_comm_info( comm, &seenRank, &seenWorld );
		fprintf( stderr, "[rank %d/%d] device %d of %d, communicator up after %.2f s (RCCL: rank %d of %d)\n", rank, world, device, nDev, since(), seenRank, seenWorld );
		if( seenRank != rank || seenWorld != world ) return 3;
		// Every rank holds the communicator, so every rank has read the id: rank 0 removes the file NOW, under an external launcher too (forked mode's
		// parent removes it again at exit) -- a launch that follows on the same port finds no id of this job, whatever its token looks like.
		if( 0 != wh_comm_barrier( comm ) ) { fprintf( stderr, "[rank %d] %s\n", rank, wh_last_error() ); return 3; }
		if( rank == 0 ) unlink( a.idFile.c_str() );

		// ---- model: rank 0 reads, everyone receives (a root that cannot read says so before the broadcast: loadModelShared) ----
		const std::wstring adapter = std::to_wstring( device ) + L":";
		sModelSetup setup;
		setup.adapter = adapter.c_str();
		iModel* model = nullptr;
		HRESULT hr = loadModelShared( widen( a.model ).c_str(), setup, nullptr, comm, 0, &model );
		if( FAILED( hr ) ) { fprintf( stderr, "[rank %d] loadModelShared failed 0x%08x\n", rank, (unsigned)hr ); return 4; }
		fprintf( stderr, "[rank %d] model ready after %.2f s\n", rank, since() );

		// ---- audio ----
		// chunk mode (default): every rank decodes THE file (16 kHz PCM is small next to the model) and takes a contiguous range of its 30 s chunks as independent
		// streams -- north_star's workload; words across chunk boundaries are cut.
		// -per-recording: whole recordings are dealt round-robin (recording i -> rank i mod world); each is ONE stream of the batch runner, i.e. the reference's own
		// host loop on it -- windows advance by the timestamp the decoder ended on (seek_delta, ContextImpl.cpp:618-625, 785) and the text of a window conditions the
		// next (prompt carry-over) --, so a recording's transcript is what whisper-main prints for it; the recordings of a rank run in lock step.
		iMediaFoundation* mf = nullptr;
		hr = initMediaFoundation( &mf );
		if( FAILED( hr ) ) return 5;
		std::vector<iAudioBuffer*> audios;
		std::vector<int> recIndex;
		std::vector<sBatchStream> streams;
		const int64_t chunk = 16000 * 30;
		int windows = 0, wb = 0, we = 0;
		if( a.perRecording )
		{
			for( size_t i = 0; i < a.wavs.size(); i++ )
			{
				if( (int)( i % (size_t)world ) != rank ) continue;
				iAudioBuffer* buf = nullptr;
				hr = mf->loadAudioFile( a.wavs[ i ].c_str(), false, &buf );
				if( FAILED( hr ) ) { fprintf( stderr, "[rank %d] cannot load %s (0x%08x)\n", rank, a.wavs[ i ].c_str(), (unsigned)hr ); return 5; }
				audios.push_back( buf );
				recIndex.push_back( (int)i );
				streams.push_back( sBatchStream{ buf, 0, (int64_t)buf->countSamples(), nullptr } );
				windows += (int)( ( (int64_t)buf->countSamples() + chunk - 1 ) / chunk );
			}
			we = windows;
		}
		else
		{
			iAudioBuffer* audio = nullptr;
			hr = mf->loadAudioFile( a.wav.c_str(), false, &audio );
			if( FAILED( hr ) ) { fprintf( stderr, "[rank %d] cannot load %s (0x%08x)\n", rank, a.wav.c_str(), (unsigned)hr ); return 5; }
			audios.push_back( audio );
			const int64_t nSamples = audio->countSamples();
			windows = (int)( ( nSamples + chunk - 1 ) / chunk );
			shardRange( windows, rank, world, wb, we );
			for( int w = wb; w < we; w++ )
				streams.push_back( sBatchStream{ audio, (int64_t)w * chunk, std::min( chunk, nSamples - (int64_t)w * chunk ), nullptr } );
		}

		sFullParams p;
		memset( &p, 0, sizeof( p ) );
		p.strategy = eSamplingStrategy::Greedy;
		p.cpuThreads = 4;
		p.n_max_text_ctx = 16384;
		p.thold_pt = p.thold_ptsum = 0.01f;
		p.beam_search.n_past = p.beam_search.beam_width = p.beam_search.n_best = -1;
		p.language = findLanguageKeyA( a.lang.c_str() );
		if( !a.perRecording ) p.setFlag( eFullParamsFlags::NoContext );	   // independent chunks carry nothing over; a whole recording keeps the reference's default
		iBatchRunner* runner = nullptr;
		const sBatchSetup bs{ (uint32_t)std::max( 1, std::min( a.slots, 512 ) ), 0, 0, 0 };
		if( !streams.empty() && FAILED( createBatchRunner( model, &bs, &runner ) ) ) return 6;

		if( 0 != wh_comm_barrier( comm ) ) { fprintf( stderr, "[rank %d] %s\n", rank, wh_last_error() ); return 8; }
		const double tStart = since();
		std::string textAll;
		int nSeg = 0;
		if( !streams.empty() )
		{
			std::vector<iTranscribeResult*> results( streams.size(), nullptr );
			hr = runner->run( p, streams.data(), (uint32_t)streams.size(), results.data(), nullptr );
			if( FAILED( hr ) ) { fprintf( stderr, "[rank %d] runFullBatch failed 0x%08x\n", rank, (unsigned)hr ); return 7; }
			for( size_t ri = 0; ri < results.size(); ri++ )
			{
				iTranscribeResult* res = results[ ri ];
				if( !res ) continue;
				std::string recText;
				std::string& text = a.perRecording ? recText : textAll;
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
				nSeg += (int)len.countSegments;
				res->Release();
				if( a.perRecording )
				{
					// one transcript per recording, written by the rank that decoded it
					FILE* rf = fopen( ( a.out + ".rec" + std::to_string( recIndex[ ri ] ) ).c_str(), "wb" );
					if( rf ) { fwrite( recText.data(), 1, recText.size(), rf ); fclose( rf ); }
				}
			}
		}
		const double tRun = since() - tStart;
		if( 0 != wh_comm_barrier( comm ) ) { fprintf( stderr, "[rank %d] %s\n", rank, wh_last_error() ); return 8; }
		const double tAll = since() - tStart;
		if( !a.perRecording )
		{
			FILE* f = fopen( ( a.out + ".rank" + std::to_string( rank ) ).c_str(), "wb" );
			if( f ) { fwrite( textAll.data(), 1, textAll.size(), f ); fclose( f ); }
		}
		fprintf( stderr, "[rank %d] chunks %d..%d of %d: %d segments in %.3f s (%.1f audio-s/s on this rank); all ranks done after %.3f s\n", rank, wb, we, windows,
			nSeg, tRun, tRun > 0 ? ( we - wb ) * 30.0 / tRun : 0.0, tAll );
		if( rank == 0 )
		{
			printf( "{\"ranks\": %d, \"windows\": %d, \"seconds\": %.4f, \"audio_seconds_per_sec\": %.2f}\n", world, windows, tAll, windows * 30.0 / tAll );
			fflush( stdout );	   // the rank leaves through _exit
		}
		if( runner ) runner->Release();
		for( iAudioBuffer* b : audios ) b->Release();
		mf->Release();
		model->Release();
		wh_comm_destroy( comm );
		return 0;
	}

	// The parent: whichever child ends first is reaped first. A non-zero exit, a signal or the deadline ends the others.
	int superviseRanks( const std::vector<pid_t>& kids, double deadlineSeconds )
	{
		const auto t0 = std::chrono::steady_clock::now();
		size_t alive = kids.size();
		int rc = 0;
		std::chrono::steady_clock::time_point tTerm;
		auto killAll = [ & ]( int sig ) { for( pid_t k : kids ) kill( k, sig ); };
		auto giveUp = [ & ]() { rc = 1; tTerm = std::chrono::steady_clock::now(); killAll( SIGTERM ); };
		while( alive > 0 )
		{
			int st = 0;
			const pid_t done = waitpid( -1, &st, WNOHANG );
			if( done > 0 )
			{
				alive--;
				const bool ok = WIFEXITED( st ) && WEXITSTATUS( st ) == 0;
				if( !ok && rc == 0 )
				{
					size_t r = 0;
					while( r < kids.size() && kids[ r ] != done ) r++;
					if( WIFEXITED( st ) ) fprintf( stderr, "whisper-mgpu: rank %zu exited with code %d; ending the other ranks\n", r, WEXITSTATUS( st ) );
					else fprintf( stderr, "whisper-mgpu: rank %zu was ended by signal %d; ending the other ranks\n", r, WIFSIGNALED( st ) ? WTERMSIG( st ) : 0 );
					giveUp();
				}
				continue;
			}
			if( done < 0 ) break;	   // no children left
			const double elapsed = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
			if( rc == 0 && deadlineSeconds > 0 && elapsed > deadlineSeconds )
			{
				fprintf( stderr, "whisper-mgpu: the job did not finish within %.0f s; ending %zu rank(s)\n", deadlineSeconds, alive );
				giveUp();
			}
			// ranks that ignore SIGTERM (stuck inside the driver) get SIGKILL two seconds later
			if( rc != 0 && std::chrono::duration<double>( std::chrono::steady_clock::now() - tTerm ).count() > 2.0 ) killAll( SIGKILL );
			std::this_thread::sleep_for( std::chrono::milliseconds( 20 ) );
		}
		return rc;
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
		else if( !strcmp( argv[ i ], "-f" ) ) { a.wav = val(); a.wavs.push_back( a.wav ); }
		else if( !strcmp( argv[ i ], "-per-recording" ) ) a.perRecording = true;
		else if( !strcmp( argv[ i ], "-l" ) ) a.lang = val();
		else if( !strcmp( argv[ i ], "-o" ) ) a.out = val();
		else if( !strcmp( argv[ i ], "-id" ) ) a.idFile = val();
		else if( !strcmp( argv[ i ], "-timeout" ) ) a.timeout = atof( val() );
		else if( !strcmp( argv[ i ], "-job-timeout" ) ) a.jobTimeout = atof( val() );
		else if( !strcmp( argv[ i ], "-job" ) ) a.job = val();
		else if( !strcmp( argv[ i ], "-slots" ) ) a.slots = atoi( val() );
		else if( !strcmp( argv[ i ], "--shard-range" ) )
		{
			// test hook (no GPU): the chunk range rank r of w gets out of n chunks, "begin end" -- must equal whisper_amd/distributed.py shard_range
			const int n = atoi( val() ), r = atoi( val() ), w = atoi( val() );
			if( n < 0 || w < 1 || r < 0 || r >= w ) return 1;
			int b = 0, e = 0;
			shardRange( n, r, w, b, e );
			printf( "%d %d\n", b, e );
			return 0;
		}
		else if( !strcmp( argv[ i ], "--id-write" ) )
		{
			// test hooks (no GPU): the id file of job <token> as rank 0 publishes it / as a rank that started <late> seconds after it takes it
			const std::string path = val(), job = val();
			unsigned char id[ WH_COMM_ID_BYTES ];
			for( size_t k = 0; k < sizeof( id ); k++ ) id[ k ] = (unsigned char)( k * 7 + 1 );
			return writeIdFile( path, job, id, sizeof( id ) ) ? 0 : 1;
		}
		else if( !strcmp( argv[ i ], "--id-read" ) )
		{
			const std::string path = val(), job = val();
			const double late = atof( val() ), timeout = atof( val() );
			unsigned char id[ WH_COMM_ID_BYTES ];
			// a rank whose own start lies `late` seconds in the future of now
			const bool ok = readIdFile( path, job, id, sizeof( id ), time( nullptr ) + (time_t)late - (time_t)timeout - 2 );
			for( size_t k = 0; ok && k < sizeof( id ); k++ )
				if( id[ k ] != (unsigned char)( k * 7 + 1 ) ) return 2;
			return ok ? 0 : 1;
		}
		else { fprintf( stderr, "usage: whisper-mgpu -n ranks -m model.bin -f audio.wav [-l en] [-o out.txt] [-timeout seconds (rendezvous and collectives)] [-job-timeout seconds (whole job; default none)] [-slots n] [-id id-file] [-job token] [-per-recording -f more.wav ...]\n" ); return 1; }
	}
	if( a.model.empty() || a.wav.empty() || a.ranks < 1 ) { fprintf( stderr, "whisper-mgpu: -m and -f are required\n" ); return 1; }
	if( !a.perRecording && a.wavs.size() > 1 ) { fprintf( stderr, "whisper-mgpu: several -f recordings need -per-recording (the chunk mode shards ONE recording)\n" ); return 1; }
	if( a.timeout <= 0 ) a.timeout = 300.0;

	// launched by torchrun / mpirun / srun: be the rank the environment names. Every rank is its own process there, so the id file
	// cannot be named after a pid: it is -id, or it is derived from the rendezvous the launcher exported.
	if( const char* wr = getenv( "RANK" ) )
	{
		const int world = getenv( "WORLD_SIZE" ) ? atoi( getenv( "WORLD_SIZE" ) ) : 1;
		const int local = getenv( "LOCAL_RANK" ) ? atoi( getenv( "LOCAL_RANK" ) ) : atoi( wr );
		if( a.idFile.empty() )
		{
			const char* port = getenv( "MASTER_PORT" );
			if( !port && world > 1 )
			{
				fprintf( stderr, "whisper-mgpu: under an external launcher (RANK is set) give every rank the same -id file, or export MASTER_PORT\n" );
				return 1;
			}
			a.idFile = std::string( "/tmp/whisper-mgpu." ) + std::to_string( (long)getuid() ) + "." + ( port ? port : "0" ) + ".id";
		}
		if( a.job.empty() )
		{
			// what every rank of ONE launch shares and two launches do not: the launcher's run / job id where it exports one, plus the rendezvous port
			for( const char* name : { "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "SLURM_STEP_ID", "OMPI_MCA_ess_base_jobid", "PMIX_NAMESPACE", "MASTER_ADDR", "MASTER_PORT" } )
				if( const char* v = getenv( name ) ) { a.job += v; a.job += '/'; }
			// A launcher that exports no per-launch id (torchrun's TORCHELASTIC_RUN_ID is the literal "none" unless --rdzv-id is given; plain mpirun exports
			// nothing of the kind) gives two launches on one port the SAME token: a rank of the second launch that starts before its rank 0 would then take the id
			// a crashed first launch left behind and block in ncclCommInitRank until -timeout. Such a job must name itself.
			const char* torchId = getenv( "TORCHELASTIC_RUN_ID" );
			const bool perLaunch = ( torchId && strcmp( torchId, "none" ) != 0 && *torchId ) || getenv( "SLURM_JOB_ID" ) || getenv( "OMPI_MCA_ess_base_jobid" ) || getenv( "PMIX_NAMESPACE" );
			if( !perLaunch && world > 1 )
			{
				fprintf( stderr, "whisper-mgpu: the launcher exports no per-launch id (TORCHELASTIC_RUN_ID is unset or 'none', no SLURM / PMIx job id): give every rank "
					"the same -job <token> that no other launch uses (e.g. -job \"$(date +%%s)-$$\" set once before the launcher), or torchrun --rdzv-id <token>\n" );
				return 1;
			}
			if( a.job.empty() ) a.job = "external";
		}
		return runRank( a, atoi( wr ), world, local );
	}
	if( a.idFile.empty() ) a.idFile = "/tmp/whisper-mgpu." + std::to_string( (long)getuid() ) + "." + std::to_string( (long)getpid() ) + ".id";
	if( a.job.empty() ) a.job = "fork/" + std::to_string( (long)getpid() ) + "/" + std::to_string( (long)time( nullptr ) );
	unlink( a.idFile.c_str() );
	std::vector<pid_t> kids;
	for( int r = 0; r < a.ranks; r++ )
	{
		const pid_t pid = fork();	// before any HIP call: a forked HIP runtime is not usable
		if( pid == 0 ) _exit( runRank( a, r, a.ranks, r ) );
		if( pid < 0 ) { perror( "fork" ); for( pid_t k : kids ) kill( k, SIGKILL ); return 1; }
		kids.push_back( pid );
	}
	// the ranks' own rendezvous and collectives give up after -timeout; the JOB has a deadline only when -job-timeout names one (a long recording or
	// a slow model read is not a hang: ADVICE r4 -- 4 x -timeout killed legitimate long jobs and deleted their partial output)
	const int rc = superviseRanks( kids, a.jobTimeout );
	unlink( a.idFile.c_str() );
	if( rc == 0 )
	{
		FILE* out = fopen( a.out.c_str(), "wb" );
		// chunk mode: the ranks' parts in rank order = the chunks in order. -per-recording: the recordings' transcripts in the order of the -f options
		const int parts = a.perRecording ? (int)a.wavs.size() : a.ranks;
		for( int r = 0; out && r < parts; r++ )
		{
			const std::string part = a.out + ( a.perRecording ? ".rec" : ".rank" ) + std::to_string( r );
			if( a.perRecording ) fprintf( out, "== %s\n", a.wavs[ (size_t)r ].c_str() );
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
	else
	{
		for( int r = 0; r < a.ranks; r++ ) unlink( ( a.out + ".rank" + std::to_string( r ) ).c_str() );
		for( size_t r = 0; r < a.wavs.size(); r++ ) unlink( ( a.out + ".rec" + std::to_string( r ) ).c_str() );
	}
	return rc;
}
