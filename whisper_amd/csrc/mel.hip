// PCM -> log-mel spectrogram on the GPU.
//
// Replaces Spectrogram::pcmToMel / SpectrogramContext::fft (Whisper/Whisper/Spectrogram.cpp:64-122,
// melSpectrogram.cpp:318-391) == log_mel_spectrogram (Whisper/source/whisper.cpp:2060-2180):
//   frames at hop 160 without centre padding (n_len = n_samples / 160), zero beyond the end; periodic Hann(400)
//   applied in FP32 like the reference; 400-point real DFT; power; the reference's fold p[j] += p[400-j], j = 1..199
//   (for a real signal p[400-j] == p[j], so the fold doubles those bins); 80x201 filterbank with a double sum;
//   log10(max(., 1e-10)) stored as float; then, over the WHOLE buffer, clamp to (max - 8) and (x + 4) / 4.
// The reference evaluates the DFT with a recursive FP32 FFT (radix-2 down to 25-point DFTs) whose own rounding noise
// is ~1e-6 of the frame energy; here the DFT is evaluated directly in FP64 against a host-built twiddle table, i.e. it
// is the exact value the reference approximates. HBM traffic is negligible (1.9 MB in, 1 MB out per 30 s); the kernel
// is LDS/FP64-issue bound and takes well under 1 % of a window's time.
#include "kernels.h"

namespace wh
{
	namespace
	{
		constexpr int N_FFT = 400, N_BINS = 201, HOP = 160;
		constexpr int FR = 8;	 // frames per workgroup

		__device__ __forceinline__ int orderedInt( float f )
		{
			const int i = __float_as_int( f );
			return i >= 0 ? i : i ^ 0x7fffffff;
		}
		__device__ __forceinline__ float fromOrderedInt( int i ) { return __int_as_float( i >= 0 ? i : i ^ 0x7fffffff ); }

		__global__ void __launch_bounds__( 256 ) melKernel( const float* __restrict__ pcm, long long nSamples,
			const float* __restrict__ filters, const double* __restrict__ dft, float* __restrict__ mel, long long nLen, int nMel,
			int* __restrict__ maxOrdered, long long nValidFrames )
		{
			__shared__ double tw[ 2 ][ N_FFT ];		  // cos, sin of 2 pi n / 400
			__shared__ double fr[ FR ][ N_FFT ];		  // windowed frames
			__shared__ double pw[ FR ][ N_BINS + 7 ];  // folded power spectrum
			__shared__ int shMax;

			const int tid = threadIdx.x;
			const long long f0 = (long long)blockIdx.x * FR;
			for( int i = tid; i < 2 * N_FFT; i += 256 ) ( &tw[ 0 ][ 0 ] )[ i ] = dft[ i ];
			if( tid == 0 ) shMax = (int)0x80000000;
			for( int i = tid; i < FR * N_FFT; i += 256 )
			{
				const int f = i / N_FFT, n = i - f * N_FFT;
				const long long s = ( f0 + f ) * HOP + n;
				// hann[n] = 0.5 * (1 - cos(2 pi n / 400)) evaluated in double and rounded to float, product in float (whisper.cpp:2073-2077, :2104)
				const float hann = (float)( 0.5 * ( 1.0 - dft[ n ] ) );
				const float x = ( s < nSamples && f0 + f < nLen ) ? pcm[ s ] : 0.0f;
				fr[ f ][ n ] = (double)( hann * x );
			}
			__syncthreads();

			// ---- DFT bins 0..200, FP64 ----
			if( tid < N_BINS )
			{
				double re[ FR ], im[ FR ];
#pragma unroll
				for( int f = 0; f < FR; f++ ) re[ f ] = im[ f ] = 0.0;
				int idx = 0;
				for( int n = 0; n < N_FFT; n++ )
				{
					const double c = tw[ 0 ][ idx ], s = tw[ 1 ][ idx ];
#pragma unroll
					for( int f = 0; f < FR; f++ )
					{
						const double x = fr[ f ][ n ];
						re[ f ] = fma( x, c, re[ f ] );
						im[ f ] = fma( -x, s, im[ f ] );
					}
					idx += tid;
					if( idx >= N_FFT ) idx -= N_FFT;
				}
				const double fold = ( tid >= 1 && tid < N_FFT / 2 ) ? 2.0 : 1.0;
#pragma unroll
				for( int f = 0; f < FR; f++ ) pw[ f ][ tid ] = fold * ( re[ f ] * re[ f ] + im[ f ] * im[ f ] );
			}
			__syncthreads();

			// ---- filterbank + log10 ----
			float localMax = -INFINITY;
			for( int o = tid; o < FR * nMel; o += 256 )
			{
				const int j = o / FR, f = o - j * FR;
				if( f0 + f >= nLen ) continue;
				const float* w = filters + (long long)j * N_BINS;
				double sum = 0.0;
				for( int kk = 0; kk < N_BINS; kk++ ) sum = fma( pw[ f ][ kk ], (double)w[ kk ], sum );
				sum = sum < 1e-10 ? 1e-10 : sum;
				// streaming: a frame the reader has no PCM chunk for is zero BEFORE normalisation (MelStreamer.cpp:229-234 memset)
				const float v = ( f0 + f < nValidFrames ) ? (float)log10( sum ) : 0.0f;
				mel[ (long long)j * nLen + f0 + f ] = v;
				localMax = fmaxf( localMax, v );
			}
			localMax = waveReduceMax( localMax );
			if( ( tid & 63 ) == 0 && localMax > -INFINITY ) atomicMax( &shMax, orderedInt( localMax ) );
			__syncthreads();
			if( tid == 0 && shMax != (int)0x80000000 ) atomicMax( maxOrdered, shMax );
		}


		// -----------------------------------------------------------------------------------------------------------
		// melKernelMf: the same arithmetic on the FP64 matrix cores (v_mfma_f64_16x16x4_f64). The direct DFT is a GEMM
		//   X[frame][bin] = sum_n a[frame][n] * tw[(n * bin) mod 400],   a = (double)( hann[n] * pcm[frame * 160 + n] )
		// and so is the filterbank  mel[frame][j] = sum_bin pw[frame][bin] * filters[j][bin]. The VALU kernel above reads one
		// LDS operand per 1.6 FMAs and runs at ~8 TFLOP/s; an MFMA reads one operand pair per 2048 FLOP.
		//   * workgroup = 16 frames (one M tile), 512 threads; the 13 tiles of 16 bins are dealt to the 8 waves, a wave keeps
		//     the real and the imaginary accumulators of its one or two tiles (A: lane & 15 = frame, lane >> 4 = sample within the 4-deep
		//     K step; B: lane & 15 = bin, lane >> 4 = sample; D: column lane & 15 = bin, row (lane >> 4) + 4 r = frame);
		//   * the 2800 raw samples the 16 overlapping frames cover sit in LDS once (11 KB), the window is applied when the
		//     operand is formed (same FP32 product as the reference), the twiddle index advances by 4 * bin mod 400 per step;
		//   * folded power spectrum [16][208] in LDS, then waves 0 .. nMel/16 - 1 run the filterbank as 51 MFMA steps each.
		constexpr int MF_FR = 16, MF_BINS = 208, MF_PW_STRIDE = 209, MF_SAMPLES = ( MF_FR - 1 ) * HOP + N_FFT;
		// sample s of the workgroup lives at s + s / 160: frames start 160 samples apart = a multiple of the 32 LDS banks, and the A
		// operand reads one sample of each of the 16 frames per instruction (16-way conflict without the skew)
		constexpr int MF_SAMPLES_SKEWED = MF_SAMPLES + MF_SAMPLES / HOP + 1;
		typedef double f64x4 __attribute__( ( ext_vector_type( 4 ) ) );

		// blockIdx.y = buffer of a batch of equally long, independent buffers (pcmStride / melStride elements apart, one maximum each): wh_mel_spectrogram_batch
		__global__ void __launch_bounds__( 512 ) melKernelMf( const float* __restrict__ pcm, long long nSamples,
			const float* __restrict__ filters, const double* __restrict__ dft, float* __restrict__ mel, long long nLen, int nMel,
			int* __restrict__ maxOrdered, long long nValidFrames, long long pcmStride = 0, long long melStride = 0 )
		{
			pcm += blockIdx.y * pcmStride;
			mel += blockIdx.y * melStride;
			maxOrdered += blockIdx.y;
			__shared__ double tw[ 2 ][ N_FFT ];
			__shared__ float hannS[ N_FFT ];
			__shared__ float pcmS[ MF_SAMPLES_SKEWED ];
			__shared__ double pw[ MF_FR ][ MF_PW_STRIDE ];
			__shared__ int shMax;

			const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
			const long long f0 = (long long)blockIdx.x * MF_FR;
			for( int i = tid; i < 2 * N_FFT; i += 512 ) ( &tw[ 0 ][ 0 ] )[ i ] = dft[ i ];
			for( int i = tid; i < N_FFT; i += 512 ) hannS[ i ] = (float)( 0.5 * ( 1.0 - dft[ i ] ) );
			if( tid == 0 ) shMax = (int)0x80000000;
			for( int i = tid; i < MF_SAMPLES; i += 512 )
			{
				const long long s = f0 * HOP + i;
				pcmS[ i + i / HOP ] = s < nSamples ? pcm[ s ] : 0.0f;
			}
			__syncthreads();

			const int row = lane & 15, kq = lane >> 4;
			// a frame at or beyond nLen is all zeros (its samples may exist: the last partial hop of the clip)
			const bool frameLive = f0 + row < nLen;
			const float* const px = pcmS + row * ( HOP + 1 ) + kq;	  // sample n of frame `row` = s = row * 160 + n -> s + row + n / 160
			// a wave owns bin tiles `wave` and `wave + 8` (13 tiles: waves 0 .. 4 have two) and runs them in ONE K loop: the A
			// operand is formed once per step and four independent accumulator chains keep the matrix core busy
			{
				const int bin0 = wave * 16 + row, bin1 = ( wave + 8 ) * 16 + row;
				const bool two = wave + 8 < MF_BINS / 16;
				int idx0 = ( kq * bin0 ) % N_FFT, idx1 = ( kq * bin1 ) % N_FFT;
				const int step0 = ( 4 * bin0 ) % N_FFT, step1 = ( 4 * bin1 ) % N_FFT;
				f64x4 re0 = { 0.0, 0.0, 0.0, 0.0 }, im0 = re0, re1 = re0, im1 = re0;
				for( int n0 = 0; n0 < N_FFT; n0 += 4 )
				{
					const int n = n0 + kq;
					const double a = frameLive ? (double)( hannS[ n ] * px[ n0 + ( n >= HOP ) + ( n >= 2 * HOP ) ] ) : 0.0;
					re0 = __builtin_amdgcn_mfma_f64_16x16x4f64( a, tw[ 0 ][ idx0 ], re0, 0, 0, 0 );
					im0 = __builtin_amdgcn_mfma_f64_16x16x4f64( a, tw[ 1 ][ idx0 ], im0, 0, 0, 0 );
					idx0 += step0;
					idx0 = idx0 >= N_FFT ? idx0 - N_FFT : idx0;
					if( two )	  // wave-uniform
					{
						re1 = __builtin_amdgcn_mfma_f64_16x16x4f64( a, tw[ 0 ][ idx1 ], re1, 0, 0, 0 );
						im1 = __builtin_amdgcn_mfma_f64_16x16x4f64( a, tw[ 1 ][ idx1 ], im1, 0, 0, 0 );
						idx1 += step1;
						idx1 = idx1 >= N_FFT ? idx1 - N_FFT : idx1;
					}
				}
				// D: column = lane & 15 = bin of the tile, row = (lane >> 4) + 4 r = frame
				const double fold0 = ( bin0 >= 1 && bin0 < N_FFT / 2 ) ? 2.0 : 1.0, fold1 = ( bin1 >= 1 && bin1 < N_FFT / 2 ) ? 2.0 : 1.0;
	#pragma unroll
				for( int r = 0; r < 4; r++ )
				{
					pw[ kq + 4 * r ][ bin0 ] = fold0 * ( re0[ r ] * re0[ r ] + im0[ r ] * im0[ r ] );
					if( two ) pw[ kq + 4 * r ][ bin1 ] = fold1 * ( re1[ r ] * re1[ r ] + im1[ r ] * im1[ r ] );
				}
			}
			__syncthreads();

			// ---- filterbank + log10: A = pw (row = frame, k = bin), B = filters (column = mel band, k = bin; zero beyond bin 200) ----
			float localMax = -INFINITY;
			if( wave * 16 < nMel )
			{
				const int j = wave * 16 + row;
				const float* const w = filters + (long long)j * N_BINS;
				f64x4 acc = { 0.0, 0.0, 0.0, 0.0 };
				for( int b0 = 0; b0 < 204; b0 += 4 )
				{
					const int b = b0 + kq;
					const double a = pw[ row ][ b ];
					const double f = b < N_BINS ? (double)w[ b ] : 0.0;
					acc = __builtin_amdgcn_mfma_f64_16x16x4f64( a, f, acc, 0, 0, 0 );
				}
	#pragma unroll
				for( int r = 0; r < 4; r++ )
				{
					const long long f = f0 + kq + 4 * r;
					if( f >= nLen ) continue;
					double sum = acc[ r ];
					sum = sum < 1e-10 ? 1e-10 : sum;
					const float v = ( f < nValidFrames ) ? (float)log10( sum ) : 0.0f;
					mel[ (long long)j * nLen + f ] = v;
					localMax = fmaxf( localMax, v );
				}
			}
			localMax = waveReduceMax( localMax );
			if( lane == 0 && localMax > -INFINITY ) atomicMax( &shMax, orderedInt( localMax ) );
			__syncthreads();
			if( tid == 0 && shMax != (int)0x80000000 ) atomicMax( maxOrdered, shMax );
		}

		__global__ void __launch_bounds__( 256 ) melNormalize( float* __restrict__ mel, long long count, const int* __restrict__ maxOrdered, long long melStride = 0 )
		{
			mel += blockIdx.y * melStride;
			const double mmax = (double)fromOrderedInt( maxOrdered[ blockIdx.y ] ) - 8.0;
			for( long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256 )
			{
				double v = (double)mel[ i ];
				v = v < mmax ? mmax : v;
				mel[ i ] = (float)( ( v + 4.0 ) / 4.0 );
			}
		}

		__global__ void melInitMax( int* maxOrdered ) { *maxOrdered = (int)0x80000000; }
		__global__ void melInitMaxN( int* maxOrdered, int n )
		{
			const int i = blockIdx.x * blockDim.x + threadIdx.x;
			if( i < n ) maxOrdered[ i ] = (int)0x80000000;
		}
		__global__ void melInitMaxFloor( int* maxOrdered, float floorValue ) { *maxOrdered = orderedInt( floorValue ); }

		// MelStreamer::makeTransposedBuffer's second pass (Whisper/Whisper/MelStreamer.cpp:148-187), all in FP32 like its SSE
		// code: mmax = max - 8; v = (max(v, mmax) + 4) * 0.25. maxOrdered[0] is this window's maximum (floor 1e-20), [1] the
		// one the previous request stored; reusePrev selects the stored one (a shorter request ending at the same frame).
		__global__ void __launch_bounds__( 256 ) melNormalizeWindow( float* __restrict__ mel, long long count, int* __restrict__ maxOrdered, int reusePrev )
		{
			const float mmax = fromOrderedInt( maxOrdered[ reusePrev ? 1 : 0 ] ) - 8.0f;
			for( long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256 )
			{
				float v = mel[ i ];
				v = v < mmax ? mmax : v;
				mel[ i ] = __fmul_rn( __fadd_rn( v, 4.0f ), 0.25f );
			}
		}
		__global__ void melKeepMax( int* maxOrdered ) { maxOrdered[ 1 ] = maxOrdered[ 0 ]; }
	}	// namespace

	// the spectrogram kernel of both entry points: FP64 matrix cores when the band count is a multiple of 16 (80, 128)
	static void launchMelFrames( const float* pcm, long long nSamples, const float* filters, const double* dftTable, float* mel, long long nLen,
		int nMel, int* mx, long long nValidFrames, hipStream_t stream )
	{
		if( ( g_tuning & TUNE_MEL_MFMA ) && ( nMel % 16 ) == 0 && nMel <= 128 )
		{
			const int blocks = (int)( ( nLen + MF_FR - 1 ) / MF_FR );
			hipLaunchKernelGGL( melKernelMf, dim3( blocks ), dim3( 512 ), 0, stream, pcm, nSamples, filters, dftTable, mel, nLen, nMel, mx, nValidFrames );
			return;
		}
		const int blocks = (int)( ( nLen + FR - 1 ) / FR );
		hipLaunchKernelGGL( melKernel, dim3( blocks ), dim3( 256 ), 0, stream, pcm, nSamples, filters, dftTable, mel, nLen, nMel, mx, nValidFrames );
	}

	int launchMel( const float* pcm, long long nSamples, const float* filters, const double* dftTable, float* mel, long long nLen,
		int nMel, float* maxScratch, hipStream_t stream )
	{
		if( nLen <= 0 ) return 0;
		int* const mx = (int*)maxScratch;
		hipLaunchKernelGGL( melInitMax, dim3( 1 ), dim3( 1 ), 0, stream, mx );
		launchMelFrames( pcm, nSamples, filters, dftTable, mel, nLen, nMel, mx, nLen, stream );
		const long long count = nLen * nMel;
		const int nb = (int)( ( count + 255 ) / 256 < 2048 ? ( count + 255 ) / 256 : 2048 );
		hipLaunchKernelGGL( melNormalize, dim3( nb ), dim3( 256 ), 0, stream, mel, count, mx );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	// `batch` independent buffers of nSamples each in THREE launches instead of 3 x batch: a 30 s window is 188 workgroups -- less than the chip -- and its three
	// launches cost ~65 us in sequence (448 windows: 29 ms of a 1.2 s batch pass). Same kernels, same arithmetic, a maximum per buffer. maxScratch: `batch` ints.
	int launchMelBatch( const float* pcm, long long nSamples, long long pcmStride, int batch, const float* filters, const double* dftTable, float* mel, long long melStride,
		long long nLen, int nMel, float* maxScratch, hipStream_t stream )
	{
		if( nLen <= 0 || batch <= 0 ) return 0;
		if( !( ( g_tuning & TUNE_MEL_MFMA ) && ( nMel % 16 ) == 0 && nMel <= 128 ) || batch > 65535 ) return 1;	   // the caller loops over launchMel
		int* const mx = (int*)maxScratch;
		hipLaunchKernelGGL( melInitMaxN, dim3( ( batch + 255 ) / 256 ), dim3( 256 ), 0, stream, mx, batch );
		const int blocks = (int)( ( nLen + MF_FR - 1 ) / MF_FR );
		hipLaunchKernelGGL( melKernelMf, dim3( blocks, batch ), dim3( 512 ), 0, stream, pcm, nSamples, filters, dftTable, mel, nLen, nMel, mx, nLen, pcmStride, melStride );
		const long long count = nLen * nMel;
		const int nb = (int)( ( count + 255 ) / 256 < 256 ? ( count + 255 ) / 256 : 256 );
		hipLaunchKernelGGL( melNormalize, dim3( nb, batch ), dim3( 256 ), 0, stream, mel, count, mx, melStride );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	// One window of a STREAMED spectrogram: frames [0, nLen) of `pcm` (the caller offsets the pointer to the window's first
	// frame), normalised by the window's own maximum -- MelStreamer::makeBuffer + makeTransposedBuffer.
	int launchMelWindow( const float* pcm, long long nSamples, const float* filters, const double* dftTable, float* mel, long long nLen,
		long long nValidFrames, int nMel, int reusePreviousMax, float* maxScratch, hipStream_t stream )
	{
		if( nLen <= 0 ) return 0;
		int* const mx = (int*)maxScratch;
		hipLaunchKernelGGL( melInitMaxFloor, dim3( 1 ), dim3( 1 ), 0, stream, mx, 1e-20f );
		launchMelFrames( pcm, nSamples, filters, dftTable, mel, nLen, nMel, mx, nValidFrames, stream );
		const long long count = nLen * nMel;
		const int nb = (int)( ( count + 255 ) / 256 < 2048 ? ( count + 255 ) / 256 : 2048 );
		hipLaunchKernelGGL( melNormalizeWindow, dim3( nb ), dim3( 256 ), 0, stream, mel, count, mx, reusePreviousMax ? 1 : 0 );
		if( !reusePreviousMax ) hipLaunchKernelGGL( melKeepMax, dim3( 1 ), dim3( 1 ), 0, stream, mx );
		WH_HIP( hipGetLastError() );
		return 0;
	}
}
