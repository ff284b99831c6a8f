// Decoder attention (self with causal mask, cross over the encoder keys) for a handful of query rows per sequence.
//
// Replaces the decoder's mulMat(K,Q) -> diagMaskInf -> softMax -> mulMat(V,.) chain of the reference
// (Whisper/Whisper/WhisperContext.cpp:455-470, 505-519; mulMatByRowTiled.hlsl, diagMaskInf.hlsl, softMax*.hlsl).
// Numerics of the reference CPU path (Whisper/source/whisper.cpp:1618-1660, 1715-1748):
//   S = K.fp16(Q) (FP32 accumulate, ggml.c:4588-4611, Q and K both pre-scaled by (d/H)^-0.25), causal -inf mask
//   (ggml.c:4967-5020), table softmax with a double sum (ggml.c:5030-5090), then P.V:
//     fast path    FP32 accumulation (what the reference's own GPU shaders do),
//     parity path  the CPU path's FP16, key-by-key, thread-partitioned accumulation (ggml.c:4689-4735 + :4615-4644)
//                  emulated exactly for `parityThreads` virtual threads.
// One 512-thread workgroup per (head, sequence, query row). HBM/latency-bound (each K/V row is read once per query
// row), so the layout of the work is chosen for bytes in flight: a thread owns whole 128-byte K rows (8 x 16-byte
// loads issued together, up to 3 rows) in the score phase and 24 x 16-byte V loads in the P.V phase; the three
// reductions (max, sum, P.V partials) go through LDS.
#include "kernels.h"

namespace wh
{
	namespace
	{
		constexpr int MAX_KEYS = 1536;
		constexpr int MAX_VTHREADS = 16;
		constexpr int NT = 512;
		constexpr int NW = NT / 64;
		constexpr int KPT = MAX_KEYS / NT;	 // keys per thread in the score phase
		constexpr int SLOTS = NT / 8;		 // key slots in the P.V phase

		__global__ void __launch_bounds__( NT ) attentionDec( const DecAttnArgs a )
		{
			__shared__ float sc[ MAX_KEYS ];
			__shared__ float qs[ HEAD_DIM ];
			__shared__ float red[ SLOTS ][ HEAD_DIM ];
			__shared__ float shf[ NW ];
			__shared__ double shd[ NW ];

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = tid >> 6;
			const int h = blockIdx.x, b = blockIdx.y, i = blockIdx.z;
			const int d = a.H * HEAD_DIM;
			const long long rowQ = (long long)b * a.nTok + i;
			const f16* const K = a.kc + ( (long long)b * a.H + h ) * a.keyStride * HEAD_DIM;
			const f16* const V = a.vc + ( (long long)b * a.H + h ) * a.keyStride * HEAD_DIM;
			int nPast = a.nPast, nKeys = a.nKeys;
			if( a.causal && a.nPastDev )
			{
				nPast = *a.nPastDev;
				nKeys = nPast + a.nTok;
			}
			// keys visible to this query row
			const int nk = a.causal ? min( nPast + i + 1, nKeys ) : nKeys;

			// ---- loads that do not wait for the position or the softmax go out first, for the first PRE = 64 keys only (a
			// decode step rarely sees more self-attention keys; cross-attention takes the dependent loads too): K row min(t, 63),
			// V row of this thread's P.V slot, and q. One memory round trip serves the common case; more keys take the
			// dependent loads below.
			constexpr int PRE = 64;
			constexpr int VPRE = 1;
			const int lastRow = a.keyStride - 1;
			const int g = tid >> 3, j8 = ( tid & 7 ) * 8;
			f16x8 kv[ KPT ][ 8 ];
			f16x8 v0[ VPRE ];
			{
				{
					int key = tid < PRE ? tid : PRE - 1;
					key = key < lastRow ? key : lastRow;
					const f16* kr = K + (long long)key * HEAD_DIM;
#pragma unroll
					for( int c8 = 0; c8 < 8; c8++ ) kv[ 0 ][ c8 ] = *(const f16x8*)( kr + c8 * 8 );
				}
				{
					const int key = g < lastRow ? g : lastRow;
					v0[ 0 ] = *(const f16x8*)( V + (long long)key * HEAD_DIM + j8 );
				}
				if( tid < HEAD_DIM ) qs[ tid ] = (float)a.q[ rowQ * d + h * HEAD_DIM + tid ];
				__syncthreads();
				// keys beyond the prefetched block (thread t owns keys t, t + 512, t + 1024)
				if( tid >= PRE && tid < nk )
				{
					const f16* kr = K + (long long)tid * HEAD_DIM;
#pragma unroll
					for( int c8 = 0; c8 < 8; c8++ ) kv[ 0 ][ c8 ] = *(const f16x8*)( kr + c8 * 8 );
				}
#pragma unroll
				for( int j = 1; j < KPT; j++ )
				{
					int key = tid + j * NT;
					key = key < nk ? key : nk - 1;
					const f16* kr = K + (long long)key * HEAD_DIM;
					if( j * NT < nk )
					{
#pragma unroll
						for( int c8 = 0; c8 < 8; c8++ ) kv[ j ][ c8 ] = *(const f16x8*)( kr + c8 * 8 );
					}
					else
					{
#pragma unroll
						for( int c8 = 0; c8 < 8; c8++ ) kv[ j ][ c8 ] = kv[ 0 ][ c8 ];
					}
				}
			}

			float mx = -INFINITY;
			float sv[ KPT ];
#pragma unroll
			for( int j = 0; j < KPT; j++ )
			{
				float s = 0.0f;
#pragma unroll
				for( int c8 = 0; c8 < 8; c8++ )
#pragma unroll
					for( int e = 0; e < 8; e++ ) s = fmaf( (float)kv[ j ][ c8 ][ e ], qs[ c8 * 8 + e ], s );
				sv[ j ] = s;
				if( tid + j * NT < nk ) mx = fmaxf( mx, s );
			}
			mx = waveReduceMax( mx );
			if( lane == 0 ) shf[ wave ] = mx;
			__syncthreads();
			mx = shf[ 0 ];
#pragma unroll
			for( int w = 1; w < NW; w++ ) mx = fmaxf( mx, shf[ w ] );

			// ---- table softmax ----
			double sum = 0.0;
#pragma unroll
			for( int j = 0; j < KPT; j++ )
			{
				const int key = tid + j * NT;
				if( key < nk )
				{
					const float e = exp16( sv[ j ] - mx );
					sv[ j ] = e;
					sum += (double)e;
				}
			}
			sum = waveReduceSumD( sum );
			if( lane == 0 ) shd[ wave ] = sum;
			__syncthreads();
			double tot = shd[ 0 ];
#pragma unroll
			for( int w = 1; w < NW; w++ ) tot += shd[ w ];
			const float inv = (float)( 1.0 / tot );
#pragma unroll
			for( int j = 0; j < KPT; j++ )
			{
				const int key = tid + j * NT;
				if( key < nk ) sc[ key ] = sv[ j ] * inv;
			}
			__syncthreads();

			float result = 0.0f;
			if( a.parityThreads <= 0 )
			{
				// ---- P.V, FP32: 64 key slots x 8 lanes of 8 dims; the first VPRE rows of the slot were prefetched above ----
				float acc[ 8 ];
#pragma unroll
				for( int j = 0; j < 8; j++ ) acc[ j ] = 0.0f;
#pragma unroll
				for( int u = 0; u < VPRE; u++ )
				{
					const int key = g + u * SLOTS;
					const float p = key < nk ? sc[ key ] : 0.0f;
#pragma unroll
					for( int j = 0; j < 8; j++ ) acc[ j ] = fmaf( (float)v0[ u ][ j ], p, acc[ j ] );
				}
				{
					if( g + SLOTS < nk )
					{
						f16x8 vv[ 7 ];
						float pp[ 7 ];
#pragma unroll
						for( int u = 0; u < 7; u++ )
						{
							const int key = g + ( u + 1 ) * SLOTS;
							const int kc = key < nk ? key : nk - 1;
							vv[ u ] = *(const f16x8*)( V + (long long)kc * HEAD_DIM + j8 );
							pp[ u ] = key < nk ? sc[ kc ] : 0.0f;
						}
#pragma unroll
						for( int u = 0; u < 7; u++ )
#pragma unroll
							for( int j = 0; j < 8; j++ ) acc[ j ] = fmaf( (float)vv[ u ][ j ], pp[ u ], acc[ j ] );
					}
					for( int k0 = g + SLOTS * 8; k0 < nk; k0 += SLOTS * 8 )
					{
						f16x8 vv[ 8 ];
						float pp[ 8 ];
#pragma unroll
						for( int u = 0; u < 8; u++ )
						{
							const int key = k0 + u * SLOTS;
							const int kc = key < nk ? key : nk - 1;
							vv[ u ] = *(const f16x8*)( V + (long long)kc * HEAD_DIM + j8 );
							pp[ u ] = key < nk ? sc[ kc ] : 0.0f;
						}
#pragma unroll
						for( int u = 0; u < 8; u++ )
#pragma unroll
							for( int j = 0; j < 8; j++ ) acc[ j ] = fmaf( (float)vv[ u ][ j ], pp[ u ], acc[ j ] );
					}
				}
#pragma unroll
				for( int j = 0; j < 8; j++ ) red[ g ][ j8 + j ] = acc[ j ];
				__syncthreads();
				if( tid < HEAD_DIM )
				{
					float t = 0.0f;
#pragma unroll
					for( int s = 0; s < SLOTS; s++ ) t += red[ s ][ tid ];
					result = t;
				}
			}
			else
			{
				// ---- P.V exactly as ggml's transposed-src0 branch: per virtual thread, y = fp16( fma( v, p, y ) ) key by key
				const int nth = min( a.parityThreads, MAX_VTHREADS );
				const int nc = nKeys;	 // the partition is over ALL key columns, masked ones contribute p = 0
				const int dc = ( nc + nth - 1 ) / nth;
				for( int vt = wave; vt < nth; vt += NW )
				{
					float y = 0.0f;
					const int k1 = min( dc * ( vt + 1 ), nc );
					for( int key = dc * vt; key < k1; key++ )
					{
						const float p = key < nk ? sc[ key ] : 0.0f;
						const float v = (float)V[ (long long)key * HEAD_DIM + lane ];
						y = round16( fmaf( v, p, y ) );
					}
					red[ vt ][ lane ] = y;
				}
				__syncthreads();
				if( tid < HEAD_DIM )
				{
					float t = red[ 0 ][ tid ];
					for( int vt = 1; vt < nth; vt++ ) t += red[ vt ][ tid ];
					result = t;
				}
			}
			if( tid < HEAD_DIM )
				a.out[ rowQ * d + h * HEAD_DIM + tid ] = (f16)result;
		}
	}	// namespace

	int launchAttentionDec( const DecAttnArgs& a, hipStream_t stream )
	{
		if( a.nKeys <= 0 || a.nKeys > MAX_KEYS || a.nTok <= 0 || a.batch <= 0 )
		{
			setError( "attentionDec: key count out of range" );
			return -1;
		}
		hipLaunchKernelGGL( attentionDec, dim3( a.H, a.batch, a.nTok ), dim3( NT ), 0, stream, a );
		WH_HIP( hipGetLastError() );
		return 0;
	}
}
