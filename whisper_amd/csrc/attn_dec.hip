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
#include <type_traits>

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
				nPast = a.nPastDev[ b ];
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

		// ---------------------------------------------------------------------------------------------------------------
		// attentionDecG: the same arithmetic, laid out for the memory system.
		//   * K rows are read the way V rows are: 8 consecutive lanes own one 128-byte row (16 bytes each), a wave instruction
		//     covers 8 whole rows = 1 KiB contiguous. The 8 partial dot products are combined with three xor-shuffles. (The
		//     first kernel gave every thread a whole row: each load instruction then touched 64 different lines for 16 bytes
		//     apiece, eight times the tag look-ups and an L1 that cannot hold the lines until their last use.)
		//   * NQ query rows share ONE pass over K and V: the hypotheses of a window in cross-attention (rows b*NQ .. b*NQ+NQ-1
		//     all attend to window b's encoder keys) -- the crossKV/b term of SURVEY.md 8(d).
		//   * FUSEQ: the cross-attention query of this head is produced here: LayerNorm of the residual row (norm.hlsl +
		//     fmaRepeat1.hlsl), product with the head's 64 rows of the query weight (L2-resident, shared by every window),
		//     bias, scale, FP16 rounding -- WhisperContext.cpp:489-501 -- instead of a LayerNorm launch and a gemv launch.
		//     The K loads of the first batch are already in flight while this runs.
		// Grid (head, window, token); 512 threads. Results differ from attentionDec only by FP32 summation order.
		constexpr int G_ROWS = NT / 8;		  // K/V rows one load instruction of the workgroup covers (64)
		constexpr int G_ITERS = MAX_KEYS / G_ROWS;	  // 24
		constexpr int G_BATCH = 8;			  // rows per thread in flight per batch
		constexpr int G_MAXD = 1280;

		template<int NQ>
		struct DecGLds
		{
			float sc[ NQ ][ MAX_KEYS ];
			float qs[ NQ ][ HEAD_DIM ];
			float red[ NQ ][ NW ][ HEAD_DIM ];
			float shf[ NQ ][ NW ];
			double shd[ NQ ][ NW ];
			f16 xn[ NQ ][ G_MAXD ];
		};

		__device__ __forceinline__ float xorReduce8( float v )
		{
			v += __shfl_xor( v, 1, 64 );
			v += __shfl_xor( v, 2, 64 );
			v += __shfl_xor( v, 4, 64 );
			return v;
		}

		// NT_LOADS: K and V rows are read with the non-temporal policy -- every row is read exactly once per launch by exactly one
		// workgroup (the cross-attention caches of a decode step: 690 MB per launch at 112 windows), so keeping it in L2 only
		// evicts what the neighbouring launches want there
		template<int NQ, bool FUSEQ, bool NT_LOADS = false>
		__global__ void __launch_bounds__( NT, 2 ) attentionDecG( const DecAttnArgs a )
		{
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemG[];
			DecGLds<NQ>& L = *(DecGLds<NQ>*)smemG;

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = tid >> 6;
			const int g = tid >> 3, c = tid & 7;
			const int h = blockIdx.x, bw = blockIdx.y, i = blockIdx.z;
			const int d = a.H * HEAD_DIM;
			const f16* const K = a.kc + ( (long long)bw * a.H + h ) * a.keyStride * HEAD_DIM;
			const f16* const V = a.vc + ( (long long)bw * a.H + h ) * a.keyStride * HEAD_DIM;
			int nPast = a.nPast, nKeys = a.nKeys;
			if( a.causal && a.nPastDev )
			{
				nPast = a.nPastDev[ bw * NQ ];	  // the sequences of a group stand at one position (hypotheses of a window)
				nKeys = nPast + a.nTok;
			}
			const int nk = a.causal ? min( nPast + i + 1, nKeys ) : nKeys;
			const int lastRow = a.keyStride - 1;
			// query row q of this workgroup
			auto rowOf = [ & ]( int q ) -> long long { return ( (long long)bw * NQ + q ) * a.nTok + i; };

			// ---- K batch 0 goes out before anything else (rows beyond nk are clamped: the position may not be known yet) ----
			f16x8 kA[ G_BATCH ], kB[ G_BATCH ];
			auto loadK = [ & ]( f16x8 ( &dst )[ G_BATCH ], int it0, int limit )
			{
	#pragma unroll
				for( int u = 0; u < G_BATCH; u++ )
				{
					int key = ( it0 + u ) * G_ROWS + g;
					key = key < limit ? key : limit;
					if constexpr( NT_LOADS )
						dst[ u ] = __builtin_nontemporal_load( (const f16x8*)( K + (long long)key * HEAD_DIM + c * 8 ) );
					else
						dst[ u ] = *(const f16x8*)( K + (long long)key * HEAD_DIM + c * 8 );
				}
			};
			loadK( kA, 0, lastRow );

			// ---- query ----
			if constexpr( FUSEQ )
			{
				// LayerNorm of the NQ residual rows, FP32 two-pass like layerNormRows (ggml.c:4098-4156 sums in double)
				const int nv = d >> 2;	  // float4 per row
				f32x4 xv[ NQ ];
				f32x4 wv = { 0, 0, 0, 0 }, bv = { 0, 0, 0, 0 };
				const bool own = tid < nv;
				if( own )
				{
					wv = *(const f32x4*)( a.lnW + tid * 4 );
					bv = *(const f32x4*)( a.lnB + tid * 4 );
				}
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
					xv[ q ] = own ? *(const f32x4*)( a.lnX + rowOf( q ) * d + tid * 4 ) : f32x4{ 0, 0, 0, 0 };
				const float invD = 1.0f / (float)d;
				float s[ NQ ];
	#pragma unroll
				for( int q = 0; q < NQ; q++ ) s[ q ] = waveReduceSum( ( xv[ q ][ 0 ] + xv[ q ][ 1 ] ) + ( xv[ q ][ 2 ] + xv[ q ][ 3 ] ) );
				if( lane == 0 )
	#pragma unroll
					for( int q = 0; q < NQ; q++ ) L.shf[ q ][ wave ] = s[ q ];
				__syncthreads();
				float mean[ NQ ];
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					float t = L.shf[ q ][ 0 ];
	#pragma unroll
					for( int w = 1; w < NW; w++ ) t += L.shf[ q ][ w ];
					mean[ q ] = t * invD;
				}
				__syncthreads();
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					float t = 0.0f;
					if( own )
					{
	#pragma unroll
						for( int e = 0; e < 4; e++ )
						{
							xv[ q ][ e ] -= mean[ q ];
							t = fmaf( xv[ q ][ e ], xv[ q ][ e ], t );
						}
					}
					s[ q ] = waveReduceSum( t );
				}
				if( lane == 0 )
	#pragma unroll
					for( int q = 0; q < NQ; q++ ) L.shf[ q ][ wave ] = s[ q ];
				__syncthreads();
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					float t = L.shf[ q ][ 0 ];
	#pragma unroll
					for( int w = 1; w < NW; w++ ) t += L.shf[ q ][ w ];
					const float rstd = 1.0f / sqrtf( t * invD + 1e-5f );
					if( own )
					{
						f16x4 hv;
	#pragma unroll
						for( int e = 0; e < 4; e++ ) hv[ e ] = (f16)__fadd_rn( __fmul_rn( __fmul_rn( xv[ q ][ e ], rstd ), wv[ e ] ), bv[ e ] );
						*(f16x4*)( &L.xn[ q ][ tid * 4 ] ) = hv;
					}
				}
				__syncthreads();
				// q[j] = fp16( ( W[h*64 + j] . xn + bias ) * scale ): 8 lanes per weight row, 128 contiguous bytes per row and step
				const f16* const wr = a.qW + ( (long long)h * HEAD_DIM + g ) * d + c * 8;
				float acc[ NQ ];
	#pragma unroll
				for( int q = 0; q < NQ; q++ ) acc[ q ] = 0.0f;
				const int steps = d >> 6;
				for( int s0 = 0; s0 < steps; s0 += 8 )
				{
					f16x8 wq[ 8 ];
	#pragma unroll
					for( int u = 0; u < 8; u++ )
						if( s0 + u < steps ) wq[ u ] = *(const f16x8*)( wr + ( s0 + u ) * 64 );
	#pragma unroll
					for( int u = 0; u < 8; u++ )
						if( s0 + u < steps )
						{
	#pragma unroll
							for( int q = 0; q < NQ; q++ )
							{
								const f16x8 xq = *(const f16x8*)( &L.xn[ q ][ ( s0 + u ) * 64 + c * 8 ] );
	#pragma unroll
								for( int e = 0; e < 8; e++ ) acc[ q ] = fmaf( (float)wq[ u ][ e ], (float)xq[ e ], acc[ q ] );
							}
						}
				}
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					const float t = xorReduce8( acc[ q ] );
					if( c == 0 ) L.qs[ q ][ g ] = round16( ( t + a.qB[ h * HEAD_DIM + g ] ) * a.qScale );
				}
			}
			else
			{
				if( tid < HEAD_DIM * NQ )
				{
					const int q = tid / HEAD_DIM, j = tid - q * HEAD_DIM;
					L.qs[ q ][ j ] = (float)a.q[ rowOf( q ) * d + h * HEAD_DIM + j ];
				}
			}
			__syncthreads();
			float qf[ NQ ][ 8 ];
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
	#pragma unroll
				for( int e = 0; e < 8; e++ ) qf[ q ][ e ] = L.qs[ q ][ c * 8 + e ];

			// ---- scores: batches of 8 row groups, the next batch in flight while this one is reduced ----
			float mx[ NQ ];
	#pragma unroll
			for( int q = 0; q < NQ; q++ ) mx[ q ] = -INFINITY;
			const int nIt = ( nk + G_ROWS - 1 ) / G_ROWS;
			auto scoreBatch = [ & ]( const f16x8 ( &kv )[ G_BATCH ], int it0 )
			{
	#pragma unroll
				for( int u = 0; u < G_BATCH; u++ )
				{
					if( it0 + u >= nIt ) break;
					const int key = ( it0 + u ) * G_ROWS + g;
	#pragma unroll
					for( int q = 0; q < NQ; q++ )
					{
						float sacc = 0.0f;
	#pragma unroll
						for( int e = 0; e < 8; e++ ) sacc = fmaf( (float)kv[ u ][ e ], qf[ q ][ e ], sacc );
						sacc = xorReduce8( sacc );
						if( key < nk )
						{
							mx[ q ] = fmaxf( mx[ q ], sacc );
							if( c == ( q & 7 ) ) L.sc[ q ][ key ] = sacc;
						}
					}
				}
			};
			// the prefetched batch 0 was clamped against the buffer, not against nk: fine, rows >= nk are ignored above
			if( nIt > G_BATCH ) loadK( kB, G_BATCH, nk - 1 );
			scoreBatch( kA, 0 );
			if( nIt > 2 * G_BATCH ) loadK( kA, 2 * G_BATCH, nk - 1 );
			if( nIt > G_BATCH ) scoreBatch( kB, G_BATCH );
			if( nIt > 2 * G_BATCH ) scoreBatch( kA, 2 * G_BATCH );

			// first V batch goes out now: it does not depend on the softmax
			f16x8 vA[ G_BATCH ], vB[ G_BATCH ];
			auto loadV = [ & ]( f16x8 ( &dst )[ G_BATCH ], int it0 )
			{
	#pragma unroll
				for( int u = 0; u < G_BATCH; u++ )
				{
					int key = ( it0 + u ) * G_ROWS + g;
					key = key < nk ? key : nk - 1;
					if constexpr( NT_LOADS )
						dst[ u ] = __builtin_nontemporal_load( (const f16x8*)( V + (long long)key * HEAD_DIM + c * 8 ) );
					else
						dst[ u ] = *(const f16x8*)( V + (long long)key * HEAD_DIM + c * 8 );
				}
			};
			const bool fast = a.parityThreads <= 0;
			if( fast ) loadV( vA, 0 );

	#pragma unroll
			for( int q = 0; q < NQ; q++ ) mx[ q ] = waveReduceMax( mx[ q ] );
			if( lane == 0 )
	#pragma unroll
				for( int q = 0; q < NQ; q++ ) L.shf[ q ][ wave ] = mx[ q ];
			__syncthreads();	// also publishes sc
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
			{
				float m = L.shf[ q ][ 0 ];
	#pragma unroll
				for( int w = 1; w < NW; w++ ) m = fmaxf( m, L.shf[ q ][ w ] );
				mx[ q ] = m;
			}

			// ---- table softmax (ggml.c:5030-5090): e = exp16( s - max ), double sum, p = e * float( 1 / sum ) ----
			double sum[ NQ ];
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
			{
				sum[ q ] = 0.0;
				for( int key = tid; key < nk; key += NT )
				{
					const float e = exp16( L.sc[ q ][ key ] - mx[ q ] );
					L.sc[ q ][ key ] = e;
					sum[ q ] += (double)e;
				}
				sum[ q ] = waveReduceSumD( sum[ q ] );
			}
			if( lane == 0 )
	#pragma unroll
				for( int q = 0; q < NQ; q++ ) L.shd[ q ][ wave ] = sum[ q ];
			__syncthreads();
			float inv[ NQ ];
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
			{
				double tot = L.shd[ q ][ 0 ];
	#pragma unroll
				for( int w = 1; w < NW; w++ ) tot += L.shd[ q ][ w ];
				inv[ q ] = (float)( 1.0 / tot );
			}

			if( fast )
			{
				// ---- P.V, FP32: slot g owns keys g, g + 64, ...; lane c owns dims c*8 .. c*8+7 ----
				float acc[ NQ ][ 8 ];
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
	#pragma unroll
					for( int e = 0; e < 8; e++ ) acc[ q ][ e ] = 0.0f;
				auto pvBatch = [ & ]( const f16x8 ( &vv )[ G_BATCH ], int it0 )
				{
	#pragma unroll
					for( int u = 0; u < G_BATCH; u++ )
					{
						const int key = ( it0 + u ) * G_ROWS + g;
						if( ( it0 + u ) >= nIt ) break;
	#pragma unroll
						for( int q = 0; q < NQ; q++ )
						{
							const float p = key < nk ? L.sc[ q ][ key ] * inv[ q ] : 0.0f;
	#pragma unroll
							for( int e = 0; e < 8; e++ ) acc[ q ][ e ] = fmaf( (float)vv[ u ][ e ], p, acc[ q ][ e ] );
						}
					}
				};
				if( nIt > G_BATCH ) loadV( vB, G_BATCH );
				pvBatch( vA, 0 );
				if( nIt > 2 * G_BATCH ) loadV( vA, 2 * G_BATCH );
				if( nIt > G_BATCH ) pvBatch( vB, G_BATCH );
				if( nIt > 2 * G_BATCH ) pvBatch( vA, 2 * G_BATCH );
				// the 8 slots of a wave first (lanes with equal c), then the 8 waves through LDS in a fixed order
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
	#pragma unroll
					for( int e = 0; e < 8; e++ )
					{
						float t = acc[ q ][ e ];
						t += __shfl_xor( t, 8, 64 );
						t += __shfl_xor( t, 16, 64 );
						t += __shfl_xor( t, 32, 64 );
						acc[ q ][ e ] = t;
					}
				if( lane < 8 )
	#pragma unroll
					for( int q = 0; q < NQ; q++ )
	#pragma unroll
						for( int e = 0; e < 8; e++ ) L.red[ q ][ wave ][ lane * 8 + e ] = acc[ q ][ e ];
				__syncthreads();
				if( tid < HEAD_DIM * NQ )
				{
					const int q = tid / HEAD_DIM, j = tid - q * HEAD_DIM;
					float t = L.red[ q ][ 0 ][ j ];
	#pragma unroll
					for( int w = 1; w < NW; w++ ) t += L.red[ q ][ w ][ j ];
					a.out[ rowOf( q ) * d + h * HEAD_DIM + j ] = (f16)t;
				}
			}
			else
			{
				// ---- P.V exactly as ggml's transposed-src0 branch (ggml.c:4689-4735 + :4615-4644), see attentionDec ----
				const int nth = min( a.parityThreads, NW );
				const int nc = nKeys;
				const int dc = ( nc + nth - 1 ) / nth;
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					__syncthreads();
					if( wave < nth )
					{
						float y = 0.0f;
						const int k1 = min( dc * ( wave + 1 ), nc );
						for( int key = dc * wave; key < k1; key++ )
						{
							const float p = key < nk ? L.sc[ q ][ key ] * inv[ q ] : 0.0f;
							const float v = (float)V[ (long long)key * HEAD_DIM + lane ];
							y = round16( fmaf( v, p, y ) );
						}
						L.red[ q ][ wave ][ lane ] = y;
					}
					__syncthreads();
					if( tid < HEAD_DIM )
					{
						float t = L.red[ q ][ 0 ][ tid ];
						for( int vt = 1; vt < nth; vt++ ) t += L.red[ q ][ vt ][ tid ];
						a.out[ rowOf( q ) * d + h * HEAD_DIM + tid ] = (f16)t;
					}
				}
			}
		}

		// ---------------------------------------------------------------------------------------------------------------
		// selfBlockDec: the self-attention half of a single-token decode step in ONE launch (WhisperContext.cpp:412-470):
		//   LayerNorm of the residual row -> this head's 64 rows of Wq, Wk, Wv (the QKV product restricted to one head)
		//   -> q = fp16((.+bq)*s), k = fp16(.*s), v = fp16(.+bv) -> k, v appended to the self-attention cache at the current
		//   position -> causal attention over the cached keys plus the new one (taken from LDS, not re-read) -> out.
		// It replaces a LayerNorm launch, a gemv launch and an attention launch. NQ sequences share one pass over the head's
		// 384 KB weight slice (L2-resident: every sequence group reads the same slice); grid (head, sequences / NQ).
		// Rounding points are those of the separate launches; only FP32 summation order differs.
		template<int NQ>
		struct SelfBlockLds
		{
			float sc[ NQ ][ MAX_KEYS / 3 ];	  // n_text_ctx <= 512 keys
			float qs[ NQ ][ HEAD_DIM ], kn[ NQ ][ HEAD_DIM ], vn[ NQ ][ HEAD_DIM ];
			float red[ NQ ][ NW ][ HEAD_DIM ];
			float shf[ NQ ][ NW ];
			double shd[ NQ ][ NW ];
			f16 xn[ NQ ][ G_MAXD ];
		};
		// MF: the projection runs on the matrix cores and needs the waves' partial tiles side by side
		template<int NQ>
		struct SelfBlockLdsMf : SelfBlockLds<NQ>
		{
			float mm[ NW ][ 12 ][ 16 ][ NQ ];	// [k share of a wave][row tile: 4 x q, 4 x k, 4 x v][row in tile][sequence]
		};

		// MF = the head's 192 weight rows x NQ activation rows as MFMA 16x16x32 tiles (the NQ rows occupy NQ of the 16 operand
		// columns): wave w takes the k-steps w, w + 8, ... of all 12 row tiles, the 8 partial tiles are added in wave order.
		// The VALU version (8 lanes per weight row, FP16 -> FP32 converts + FMAs) is bound by those converts: ~2500 VALU
		// instructions per lane at NQ = 4, d = 1024.
		template<int NQ, bool MF>
		__global__ void __launch_bounds__( NT, NQ > 4 ? 1 : 2 ) selfBlockDec( const DecSelfArgs a )
		{
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemS[];
			using Lds = std::conditional_t<MF, SelfBlockLdsMf<NQ>, SelfBlockLds<NQ>>;
			Lds& L = *(Lds*)smemS;
			const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
			const int g = tid >> 3, c = tid & 7;
			const int h = blockIdx.x, sg = blockIdx.y;
			const int d = a.H * HEAD_DIM;
			const int nSeq = min( NQ, a.batch - sg * NQ );
			auto seqOf = [ & ]( int q ) { return sg * NQ + ( q < nSeq ? q : nSeq - 1 ); };
			// position of the token being fed = number of cached keys, PER SEQUENCE: the sequences of a lock-step batch may carry prompts of
			// different lengths (a batch scheduler's streams). Loops run to the group's longest; a sequence masks what lies beyond its own.
			int pos[ NQ ];
			int posMax = 0;
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
			{
				pos[ q ] = a.nPastDev ? a.nPastDev[ seqOf( q ) ] : a.nPast;
				posMax = max( posMax, pos[ q ] );
			}
			auto cacheOf = [ & ]( const f16* base, int q ) { return base + ( (long long)seqOf( q ) * a.H + h ) * a.keyStride * HEAD_DIM; };

			// cached K rows 0..63 of every sequence go out first (clamped against the buffer: rows >= pos are ignored later)
			f16x8 k0[ NQ ];
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
				k0[ q ] = *(const f16x8*)( cacheOf( a.kc, q ) + (long long)min( g, a.keyStride - 1 ) * HEAD_DIM + c * 8 );

			// ---- LayerNorm of the NQ residual rows (same arithmetic as attentionDecG's fused query) ----
			{
				const int nv = d >> 2;
				f32x4 xv[ NQ ];
				f32x4 wv = { 0, 0, 0, 0 }, bv = { 0, 0, 0, 0 };
				const bool own = tid < nv;
				if( own )
				{
					wv = *(const f32x4*)( a.lnW + tid * 4 );
					bv = *(const f32x4*)( a.lnB + tid * 4 );
				}
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
					xv[ q ] = own ? *(const f32x4*)( a.x + (long long)seqOf( q ) * d + tid * 4 ) : f32x4{ 0, 0, 0, 0 };
				const float invD = 1.0f / (float)d;
				float s[ NQ ];
	#pragma unroll
				for( int q = 0; q < NQ; q++ ) s[ q ] = waveReduceSum( ( xv[ q ][ 0 ] + xv[ q ][ 1 ] ) + ( xv[ q ][ 2 ] + xv[ q ][ 3 ] ) );
				if( lane == 0 )
	#pragma unroll
					for( int q = 0; q < NQ; q++ ) L.shf[ q ][ wave ] = s[ q ];
				__syncthreads();
				float mean[ NQ ];
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					float t = L.shf[ q ][ 0 ];
	#pragma unroll
					for( int w = 1; w < NW; w++ ) t += L.shf[ q ][ w ];
					mean[ q ] = t * invD;
				}
				__syncthreads();
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					float t = 0.0f;
					if( own )
					{
	#pragma unroll
						for( int e = 0; e < 4; e++ )
						{
							xv[ q ][ e ] -= mean[ q ];
							t = fmaf( xv[ q ][ e ], xv[ q ][ e ], t );
						}
					}
					s[ q ] = waveReduceSum( t );
				}
				if( lane == 0 )
	#pragma unroll
					for( int q = 0; q < NQ; q++ ) L.shf[ q ][ wave ] = s[ q ];
				__syncthreads();
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					float t = L.shf[ q ][ 0 ];
	#pragma unroll
					for( int w = 1; w < NW; w++ ) t += L.shf[ q ][ w ];
					const float rstd = 1.0f / sqrtf( t * invD + 1e-5f );
					if( own )
					{
						f16x4 hv;
	#pragma unroll
						for( int e = 0; e < 4; e++ ) hv[ e ] = (f16)__fadd_rn( __fmul_rn( __fmul_rn( xv[ q ][ e ], rstd ), wv[ e ] ), bv[ e ] );
						*(f16x4*)( &L.xn[ q ][ tid * 4 ] ) = hv;
					}
				}
				__syncthreads();
			}

			if constexpr( MF )
			{
				// ---- this head's rows of Wq, Wk, Wv on the matrix cores ----
				// A = 16 weight rows x 32 k (lane & 15 = row, lane >> 4 = 8-half chunk), B = the NQ normalised rows from LDS in
				// operand columns 0 .. NQ-1 (the other columns repeat the last row; their results are never read),
				// D[row = (lane >> 4) * 4 + r][col = lane & 15]
				const int steps = d >> 5;
				const int qCol = min( lane & 15, NQ - 1 );
				const f16* const xrow = &L.xn[ qCol ][ ( lane >> 4 ) * 8 ];
				const f16* const wbase = a.wqkv + ( (long long)h * HEAD_DIM + ( lane & 15 ) ) * d + ( lane >> 4 ) * 8;
	#pragma unroll
				for( int m = 0; m < 3; m++ )
				{
					f32x4 acc4[ 4 ];
	#pragma unroll
					for( int t = 0; t < 4; t++ ) acc4[ t ] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
					const f16* const wm = wbase + (long long)m * d * d;
					constexpr int U = 2;	  // k-steps in flight per wave (4 keeps the kernel above 128 registers: one workgroup per CU)
					for( int s0 = wave; s0 < steps; s0 += U * NW )
					{
						f16x8 wf[ U ][ 4 ], xf[ U ];
	#pragma unroll
						for( int u = 0; u < U; u++ )
							if( s0 + u * NW < steps )
							{
	#pragma unroll
								for( int t = 0; t < 4; t++ ) wf[ u ][ t ] = *(const f16x8*)( wm + (long long)t * 16 * d + ( s0 + u * NW ) * 32 );
								xf[ u ] = *(const f16x8*)( xrow + ( s0 + u * NW ) * 32 );
							}
	#pragma unroll
						for( int u = 0; u < U; u++ )
							if( s0 + u * NW < steps )
							{
	#pragma unroll
								for( int t = 0; t < 4; t++ ) acc4[ t ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( wf[ u ][ t ], xf[ u ], acc4[ t ], 0, 0, 0 );
							}
					}
					if( ( lane & 15 ) < NQ )
	#pragma unroll
						for( int t = 0; t < 4; t++ )
	#pragma unroll
							for( int r = 0; r < 4; r++ ) L.mm[ wave ][ m * 4 + t ][ ( lane >> 4 ) * 4 + r ][ lane & 15 ] = acc4[ t ][ r ];
				}
				__syncthreads();
				// lane c < NQ of weight row g finishes sequence c: the 8 k-shares in wave order, then the reference's rounding points
				if( c < NQ )
				{
					const int q = c;
					float tq = 0.0f, tk = 0.0f, tv = 0.0f;
	#pragma unroll
					for( int w = 0; w < NW; w++ )
					{
						tq += L.mm[ w ][ 0 + ( g >> 4 ) ][ g & 15 ][ q ];
						tk += L.mm[ w ][ 4 + ( g >> 4 ) ][ g & 15 ][ q ];
						tv += L.mm[ w ][ 8 + ( g >> 4 ) ][ g & 15 ][ q ];
					}
					const int col = h * HEAD_DIM + g;
					const float bq = a.bqkv[ col ], bv = a.bqkv[ 2 * d + col ];
					const f16 hq = (f16)( ( tq + bq ) * a.scale ), hk = (f16)( tk * a.scale ), hv = (f16)( tv + bv );
					L.qs[ q ][ g ] = (float)hq;
					L.kn[ q ][ g ] = (float)hk;
					L.vn[ q ][ g ] = (float)hv;
					if( q < nSeq )
					{
						const long long o = ( ( (long long)seqOf( q ) * a.H + h ) * a.keyStride + pos[ q ] ) * HEAD_DIM + g;
						a.kc[ o ] = hk;
						a.vc[ o ] = hv;
					}
				}
			}
			else
			// ---- this head's rows of Wq, Wk, Wv: 8 lanes per weight row, 128 contiguous bytes per row and step ----
			{
				float acc[ 3 ][ NQ ];
	#pragma unroll
				for( int m = 0; m < 3; m++ )
	#pragma unroll
					for( int q = 0; q < NQ; q++ ) acc[ m ][ q ] = 0.0f;
				const f16* wr[ 3 ];
	#pragma unroll
				for( int m = 0; m < 3; m++ ) wr[ m ] = a.wqkv + ( (long long)m * d + h * HEAD_DIM + g ) * d + c * 8;
				const int steps = d >> 6;
				for( int s0 = 0; s0 < steps; s0 += 8 )
				{
					f16x8 wq[ 3 ][ 8 ];
	#pragma unroll
					for( int m = 0; m < 3; m++ )
	#pragma unroll
						for( int u = 0; u < 8; u++ )
							if( s0 + u < steps ) wq[ m ][ u ] = *(const f16x8*)( wr[ m ] + ( s0 + u ) * 64 );
	#pragma unroll
					for( int u = 0; u < 8; u++ )
						if( s0 + u < steps )
						{
	#pragma unroll
							for( int q = 0; q < NQ; q++ )
							{
								const f16x8 xq = *(const f16x8*)( &L.xn[ q ][ ( s0 + u ) * 64 + c * 8 ] );
	#pragma unroll
								for( int m = 0; m < 3; m++ )
	#pragma unroll
									for( int e = 0; e < 8; e++ ) acc[ m ][ q ] = fmaf( (float)wq[ m ][ u ][ e ], (float)xq[ e ], acc[ m ][ q ] );
							}
						}
				}
				const int col = h * HEAD_DIM + g;
				const float bq = a.bqkv[ col ], bv = a.bqkv[ 2 * d + col ];
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					const float tq = xorReduce8( acc[ 0 ][ q ] ), tk = xorReduce8( acc[ 1 ][ q ] ), tv = xorReduce8( acc[ 2 ][ q ] );
					if( c == 0 )
					{
						const f16 hq = (f16)( ( tq + bq ) * a.scale ), hk = (f16)( tk * a.scale ), hv = (f16)( tv + bv );
						L.qs[ q ][ g ] = (float)hq;
						L.kn[ q ][ g ] = (float)hk;
						L.vn[ q ][ g ] = (float)hv;
						if( q < nSeq )
						{
							const long long o = ( ( (long long)seqOf( q ) * a.H + h ) * a.keyStride + pos[ q ] ) * HEAD_DIM + g;
							a.kc[ o ] = hk;
							a.vc[ o ] = hv;
						}
					}
				}
			}
			__syncthreads();

			// ---- scores: cached keys (8 lanes per row) + the new key from LDS ----
			float qf[ NQ ][ 8 ];
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
	#pragma unroll
				for( int e = 0; e < 8; e++ ) qf[ q ][ e ] = L.qs[ q ][ c * 8 + e ];
			float mx[ NQ ];
	#pragma unroll
			for( int q = 0; q < NQ; q++ ) mx[ q ] = -INFINITY;
			const int nIt = ( posMax + G_ROWS - 1 ) / G_ROWS;
			for( int it = 0; it < nIt; it++ )
			{
				const int key = it * G_ROWS + g;
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					const f16x8 kv = it == 0 ? k0[ q ] : *(const f16x8*)( cacheOf( a.kc, q ) + (long long)max( min( key, pos[ q ] - 1 ), 0 ) * HEAD_DIM + c * 8 );
					float sacc = 0.0f;
	#pragma unroll
					for( int e = 0; e < 8; e++ ) sacc = fmaf( (float)kv[ e ], qf[ q ][ e ], sacc );
					sacc = xorReduce8( sacc );
					if( key < pos[ q ] )
					{
						mx[ q ] = fmaxf( mx[ q ], sacc );
						if( c == ( q & 7 ) ) L.sc[ q ][ key ] = sacc;
					}
				}
			}
			// first V group of the cache goes out before the softmax
			f16x8 v0[ NQ ];
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
				v0[ q ] = *(const f16x8*)( cacheOf( a.vc, q ) + (long long)min( g, max( pos[ q ] - 1, 0 ) ) * HEAD_DIM + c * 8 );
			if( wave == 0 )
			{
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					const float sNew = waveReduceSum( L.qs[ q ][ lane ] * L.kn[ q ][ lane ] );
					mx[ q ] = fmaxf( mx[ q ], sNew );
					if( lane == 0 ) L.sc[ q ][ pos[ q ] ] = sNew;
				}
			}
	#pragma unroll
			for( int q = 0; q < NQ; q++ ) mx[ q ] = waveReduceMax( mx[ q ] );
			if( lane == 0 )
	#pragma unroll
				for( int q = 0; q < NQ; q++ ) L.shf[ q ][ wave ] = mx[ q ];
			__syncthreads();
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
			{
				float m = L.shf[ q ][ 0 ];
	#pragma unroll
				for( int w = 1; w < NW; w++ ) m = fmaxf( m, L.shf[ q ][ w ] );
				mx[ q ] = m;
			}
			double sum[ NQ ];
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
			{
				sum[ q ] = 0.0;
				for( int key = tid; key <= pos[ q ]; key += NT )
				{
					const float e = exp16( L.sc[ q ][ key ] - mx[ q ] );
					L.sc[ q ][ key ] = e;
					sum[ q ] += (double)e;
				}
				sum[ q ] = waveReduceSumD( sum[ q ] );
			}
			if( lane == 0 )
	#pragma unroll
				for( int q = 0; q < NQ; q++ ) L.shd[ q ][ wave ] = sum[ q ];
			__syncthreads();
			float inv[ NQ ];
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
			{
				double tot = L.shd[ q ][ 0 ];
	#pragma unroll
				for( int w = 1; w < NW; w++ ) tot += L.shd[ q ][ w ];
				inv[ q ] = (float)( 1.0 / tot );
			}

			// ---- P.V over the cached rows, FP32; the new row is added from LDS at the end ----
			float acc[ NQ ][ 8 ];
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
	#pragma unroll
				for( int e = 0; e < 8; e++ ) acc[ q ][ e ] = 0.0f;
			for( int it = 0; it < nIt; it++ )
			{
				const int key = it * G_ROWS + g;
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
				{
					const f16x8 vv = it == 0 ? v0[ q ] : *(const f16x8*)( cacheOf( a.vc, q ) + (long long)max( min( key, pos[ q ] - 1 ), 0 ) * HEAD_DIM + c * 8 );
					const float p = key < pos[ q ] ? L.sc[ q ][ key ] * inv[ q ] : 0.0f;
	#pragma unroll
					for( int e = 0; e < 8; e++ ) acc[ q ][ e ] = fmaf( (float)vv[ e ], p, acc[ q ][ e ] );
				}
			}
	#pragma unroll
			for( int q = 0; q < NQ; q++ )
	#pragma unroll
				for( int e = 0; e < 8; e++ )
				{
					float t = acc[ q ][ e ];
					t += __shfl_xor( t, 8, 64 );
					t += __shfl_xor( t, 16, 64 );
					t += __shfl_xor( t, 32, 64 );
					acc[ q ][ e ] = t;
				}
			if( lane < 8 )
	#pragma unroll
				for( int q = 0; q < NQ; q++ )
	#pragma unroll
					for( int e = 0; e < 8; e++ ) L.red[ q ][ wave ][ lane * 8 + e ] = acc[ q ][ e ];
			__syncthreads();
			if( tid < HEAD_DIM * NQ )
			{
				const int q = tid / HEAD_DIM, j = tid - q * HEAD_DIM;
				if( q < nSeq )
				{
					float t = L.red[ q ][ 0 ][ j ];
	#pragma unroll
					for( int w = 1; w < NW; w++ ) t += L.red[ q ][ w ][ j ];
					t = fmaf( L.vn[ q ][ j ], L.sc[ q ][ pos[ q ] ] * inv[ q ], t );
					a.out[ (long long)seqOf( q ) * d + h * HEAD_DIM + j ] = (f16)t;
				}
			}
		}

		// ---------------------------------------------------------------------------------------------------------------
		// selfAttnDecWave: causal self-attention of a single-token decode step for MANY sequences (a lock-step batch of more than 128: there the
		// fused selfBlockDec no longer fits one round of workgroups and the self-attention half runs as its own launch behind the QKV product).
		// A WAVE per (sequence, head) -- a sequence sees at most n_text_ctx <= 512 keys, 7 KB of K and V each at 55 keys -- four of them per
		// workgroup, no workgroup barrier. Scores: a lane owns whole K rows (keys lane, lane + 64, ...; the FMAs in attentionDec's order, so the
		// scores are the same bits); the reference's table softmax with a double sum; P.V in FP32: 8 key slots x 8 lanes of 8 dims, slots added by
		// shuffles. The new token's K / V rows were appended to the cache by the QKV product's epilogue (EPI_QKV_DEC) before this launch.
		constexpr int SAW_WAVES = 4, SAW_MAX_KEYS = 512;
		__global__ void __launch_bounds__( SAW_WAVES * 64 ) selfAttnDecWave( const DecAttnArgs a )
		{
			__shared__ float scW[ SAW_WAVES ][ SAW_MAX_KEYS ];
			__shared__ float qsW[ SAW_WAVES ][ HEAD_DIM ];
			const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
			const int pair = blockIdx.x * SAW_WAVES + wave;
			if( pair >= a.batch * a.H ) return;
			const int b = pair / a.H, h = pair - b * a.H;
			const int d = a.H * HEAD_DIM;
			const int nPast = a.nPastDev ? a.nPastDev[ b ] : a.nPast;
			const int nk = min( nPast + 1, a.keyStride );
			const f16* const K = a.kc + ( (long long)b * a.H + h ) * a.keyStride * HEAD_DIM;
			const f16* const V = a.vc + ( (long long)b * a.H + h ) * a.keyStride * HEAD_DIM;
			float* const sc = scW[ wave ];
			float* const qs = qsW[ wave ];
			// the first 64 K rows and the first 8 V rows go out before anything else (a decode step rarely sees more keys)
			f16x8 k0[ 8 ];
			{
				const f16* kr = K + (long long)min( lane, nk - 1 ) * HEAD_DIM;
#pragma unroll
				for( int c8 = 0; c8 < 8; c8++ ) k0[ c8 ] = *(const f16x8*)( kr + c8 * 8 );
			}
			const int g = lane >> 3, j8 = ( lane & 7 ) * 8;
			f16x8 v0 = *(const f16x8*)( V + (long long)min( g, nk - 1 ) * HEAD_DIM + j8 );
			qs[ lane ] = (float)a.q[ (long long)b * d + h * HEAD_DIM + lane ];
			__builtin_amdgcn_wave_barrier();	   // LDS operations of a wave execute in order; this only keeps the compiler from moving reads above the write
			// ---- scores ----
			float mx = -INFINITY;
			for( int k00 = 0; k00 < nk; k00 += 64 )
			{
				const int key = k00 + lane;
				if( k00 > 0 )
				{
					const f16* kr = K + (long long)min( key, nk - 1 ) * HEAD_DIM;
#pragma unroll
					for( int c8 = 0; c8 < 8; c8++ ) k0[ c8 ] = *(const f16x8*)( kr + c8 * 8 );
				}
				float sAcc = 0.0f;
#pragma unroll
				for( int c8 = 0; c8 < 8; c8++ )
#pragma unroll
					for( int e = 0; e < 8; e++ ) sAcc = fmaf( (float)k0[ c8 ][ e ], qs[ c8 * 8 + e ], sAcc );
				if( key < nk )
				{
					sc[ key ] = sAcc;
					mx = fmaxf( mx, sAcc );
				}
			}
			mx = waveReduceMax( mx );
			__builtin_amdgcn_wave_barrier();
			// ---- table softmax (ggml.c:5030-5090): e = exp16( s - max ), double sum, p = e * float( 1 / sum ) ----
			double sum = 0.0;
			for( int key = lane; key < nk; key += 64 )
			{
				const float e = exp16( sc[ key ] - mx );
				sc[ key ] = e;
				sum += (double)e;
			}
			sum = waveReduceSumD( sum );
			__builtin_amdgcn_wave_barrier();
			const float inv = (float)( 1.0 / sum );
			// ---- P.V ----
			float acc[ 8 ];
#pragma unroll
			for( int j = 0; j < 8; j++ ) acc[ j ] = 0.0f;
			{
				const float p = g < nk ? sc[ g ] * inv : 0.0f;
#pragma unroll
				for( int j = 0; j < 8; j++ ) acc[ j ] = fmaf( (float)v0[ j ], p, acc[ j ] );
			}
			for( int k00 = 8; k00 < nk; k00 += 64 )
			{
				f16x8 vv[ 8 ];
				float pp[ 8 ];
#pragma unroll
				for( int u = 0; u < 8; u++ )
				{
					const int key = k00 + u * 8 + g;
					const int kcl = min( key, nk - 1 );
					vv[ u ] = *(const f16x8*)( V + (long long)kcl * HEAD_DIM + j8 );
					pp[ u ] = key < nk ? sc[ kcl ] * inv : 0.0f;
				}
#pragma unroll
				for( int u = 0; u < 8; u++ )
#pragma unroll
					for( int j = 0; j < 8; j++ ) acc[ j ] = fmaf( (float)vv[ u ][ j ], pp[ u ], acc[ j ] );
			}
#pragma unroll
			for( int j = 0; j < 8; j++ )
			{
				float t = acc[ j ];
				t += __shfl_xor( t, 8, 64 );
				t += __shfl_xor( t, 16, 64 );
				t += __shfl_xor( t, 32, 64 );
				acc[ j ] = t;
			}
			if( lane < 8 )
			{
				f16x8 o;
#pragma unroll
				for( int j = 0; j < 8; j++ ) o[ j ] = (f16)acc[ j ];
				*(f16x8*)( a.out + (long long)b * d + h * HEAD_DIM + lane * 8 ) = o;
			}
		}

		template<int NQ, bool MF>
		int launchSelfBlockK( const DecSelfArgs& a, hipStream_t stream )
		{
			constexpr int lds = (int)sizeof( std::conditional_t<MF, SelfBlockLdsMf<NQ>, SelfBlockLds<NQ>> );
			static_assert( lds <= 160 * 1024, "selfBlockDec LDS" );
			if( lds > 48 * 1024 )
			{
				static PerDeviceOnce once;
				if( const int onceDev = once.needed(); onceDev >= 0 )
				{
					WH_HIP( hipFuncSetAttribute( (const void*)selfBlockDec<NQ, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds ) );
					once.mark( onceDev );
				}
			}
			hipLaunchKernelGGL( ( selfBlockDec<NQ, MF> ), dim3( a.H, ( a.batch + NQ - 1 ) / NQ ), dim3( NT ), lds, stream, a );
			WH_HIP( hipGetLastError() );
			return 0;
		}

		// ---------------------------------------------------------------------------------------------------------------
		// attentionDecM (round 6): the cross-attention of a decode step for HYPOTHESIS GROUPS -- NQ = 2 .. 8 rows (the hypotheses of a window in beam search) share
		// one pass over the window's keys -- on the matrix cores. attentionDecG<5, true> spends 48 us per launch at 8 windows x 20 heads (1.3 TB/s): ~4500 VALU
		// instructions per lane (8-lane partial dot products, three shuffles each, for 5 queries per key row), a fused query projection of 8 lanes per weight row, and
		// a chain of ten dependent memory round trips. Here, per (head, window) workgroup of 8 waves:
		//   * every load of the kernel's first half is in flight before anything is computed: the head's 64 query-weight rows (MFMA operand fragments, k-steps
		//     dealt to the waves) and ALL K tiles of the wave (tiles wave, wave + 8, ...: 16 keys x 64 dims each, two 16-byte fragments per lane);
		//   * LayerNorm of row q by wave q (layerNormRows: the bits of the standalone kernel), query projection as 4 row tiles x d / 32 k-steps of
		//     v_mfma_f32_16x16x32_f16 with the NQ rows in NQ of the 16 operand columns, the 8 partial tiles added in wave order;
		//   * S^T = K . Q^T: two MFMAs per tile of 16 keys; a lane ends up with 4 consecutive keys of ONE query (column lane & 15) per tile -- the scores stay
		//     in registers (12 tiles x 4), maxima meet through LDS, e = exp16( s - max ), double sums (ggml.c:5030-5090);
		//   * O^T = V^T . P^T: P's MFMA operand IS the lane's own e registers (k index = 8 (lane >> 4) + e <-> tile t0 / t1, key 4 (lane >> 4) + (e & 3)); V is
		//     stored [key][64] (attentionDecG's layout), so its operand needs 8 keys of one dim per lane: each wave stages the two V tiles of a 32-key block
		//     in its own 4 KiB of LDS (coalesced 128-byte rows in, chunks XOR-swizzled by row / 4, conflict-free 2-byte reads out);
		//   * e is an FP16 value, so P enters the product exactly; the 1 / sum factor is applied to the FP32 result (the reference rounds e / sum to FP16 first,
		//     ggml.c:5912-6097; attentionDecG multiplies in FP32): results differ from attentionDecG<NQ> by summation order and that one rounding.
		// Grid (head, window); nTok == 1, not causal, fused query only.
		constexpr int M_TPW = MAX_KEYS / 16 / NW;	  // 12 key tiles per wave
		constexpr int M_QSTEPS = G_MAXD / 32 / NW;	  // 5 k-steps of the query projection per wave

		template<int NQ>
		struct DecMLds
		{
			float part[ NW ][ 4 ][ 4 ][ 64 ];		   // the waves' partial tiles [wave][tile][register][lane]: query projection, later O^T
			unsigned char vst[ NW ][ 2 ][ 2048 ];	   // per wave: the two V tiles of a 32-key block
			f16 xn[ NQ ][ G_MAXD ];
			f16 qh[ NQ ][ HEAD_DIM ];
			float wmax[ NW ][ 16 ];
			double wsum[ NW ][ 16 ];
		};

		// (the ablation instances behind `cross_ablate` -- no exponentials / no V transposes / no V loads / no K loads -- were measured and removed again: the numbers are in
		// profiles/r06_evidence/small_batch_products.txt, the code in the history at commit 1576939, the records of session r6v)
		template<int NQ>
		__global__ void __launch_bounds__( NT, 1 ) attentionDecM( const DecAttnArgs a )
		{
			static_assert( NQ >= 1 && NQ <= NW, "wave q normalises row q" );
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemM[];
			DecMLds<NQ>& L = *(DecMLds<NQ>*)smemM;
			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
			const int lr = lane & 15, lg = lane >> 4;
			const int h = blockIdx.x, bw = blockIdx.y;
			const int d = a.H * HEAD_DIM;
			const f16* const K = a.kc + ( (long long)bw * a.H + h ) * a.keyStride * HEAD_DIM;
			const f16* const V = a.vc + ( (long long)bw * a.H + h ) * a.keyStride * HEAD_DIM;
			const int nk = a.nKeys;
			const int nTiles = ( nk + 15 ) >> 4;
			const int steps = d >> 5;

			// ---- everything the first half needs goes out now: query weights (row tile rt, k-step wave + 8 s), then the wave's K tiles ----
			// (measured and not kept: the residual row's loads in front of these and all twelve V tiles requested before the scores -- 28.5 against 26.8 us)
			f16x8 wq[ M_QSTEPS ][ 4 ];
	#pragma unroll
			for( int s = 0; s < M_QSTEPS; s++ )
			{
				const int ks = wave + NW * s;
				if( ks < steps )
	#pragma unroll
					for( int rt = 0; rt < 4; rt++ ) wq[ s ][ rt ] = *(const f16x8*)( a.qW + ( (long long)h * HEAD_DIM + rt * 16 + lr ) * d + ks * 32 + lg * 8 );
			}
			f16x8 kf[ M_TPW ][ 2 ];
	#pragma unroll
			for( int j = 0; j < M_TPW; j++ )
			{
				const int t = wave + NW * j;
				if( t < nTiles )
				{
					int key = t * 16 + lr;
					key = key < nk ? key : nk - 1;
					const f16* const p = K + (long long)key * HEAD_DIM + lg * 8;
					kf[ j ][ 0 ] = __builtin_nontemporal_load( (const f16x8*)p );
					kf[ j ][ 1 ] = __builtin_nontemporal_load( (const f16x8*)( p + 32 ) );
				}
			}

			// ---- LayerNorm of the group's residual rows: wave q takes row q ----
			if( wave < NQ )
				layerNormRows<G_MAXD / 256, 1>( a.lnX + ( (long long)bw * NQ + wave ) * d, 0, 1, a.lnW, a.lnB, d, lane, [ & ]( int, int c, f16x4 v ) { *(f16x4*)( &L.xn[ wave ][ c ] ) = v; } );
			__syncthreads();

			// ---- query projection: D[row = weight row of the tile][col = q] ----
			const int qrow = lr < NQ ? lr : NQ - 1;	   // operand columns beyond the group repeat its last row (their results are never read)
			{
				f32x4 qa[ 4 ];
	#pragma unroll
				for( int rt = 0; rt < 4; rt++ ) qa[ rt ] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
	#pragma unroll
				for( int s = 0; s < M_QSTEPS; s++ )
				{
					const int ks = wave + NW * s;
					if( ks < steps )
					{
						const f16x8 xb = *(const f16x8*)( &L.xn[ qrow ][ ks * 32 + lg * 8 ] );
	#pragma unroll
						for( int rt = 0; rt < 4; rt++ ) qa[ rt ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( wq[ s ][ rt ], xb, qa[ rt ], 0, 0, 0 );
					}
				}
	#pragma unroll
				for( int rt = 0; rt < 4; rt++ )
	#pragma unroll
					for( int r = 0; r < 4; r++ ) L.part[ wave ][ rt ][ r ][ lane ] = qa[ rt ][ r ];
			}
			__syncthreads();
			if( tid < NQ * HEAD_DIM )
			{
				const int q = tid >> 6, j = tid & 63;
				const int rt = j >> 4, ln = ( ( j & 15 ) >> 2 ) * 16 + q, r = j & 3;
				float t = L.part[ 0 ][ rt ][ r ][ ln ];
	#pragma unroll
				for( int w = 1; w < NW; w++ ) t += L.part[ w ][ rt ][ r ][ ln ];
				L.qh[ q ][ j ] = (f16)( ( t + a.qB[ h * HEAD_DIM + j ] ) * a.qScale );
			}
			__syncthreads();

			// ---- the first half of the wave's V tiles goes out before the scores wait for K (lane = row lane >> 3 (+ 8), chunk lane & 7: whole 128-byte rows) ----
			f16x8 vf[ M_TPW ][ 2 ];
			auto loadV = [ & ]( int j )
			{
				const int t = wave + NW * j;
	#pragma unroll
				for( int i = 0; i < 2; i++ )
				{
					int key = t * 16 + ( lane >> 3 ) + 8 * i;
					key = key < nk ? key : nk - 1;
					vf[ j ][ i ] = __builtin_nontemporal_load( (const f16x8*)( V + (long long)key * HEAD_DIM + ( lane & 7 ) * 8 ) );
				}
			};
	#pragma unroll
			for( int j = 0; j < M_TPW / 2; j++ )
				if( wave + NW * ( j & ~1 ) < nTiles ) loadV( j );

			// ---- scores: S^T[key][q], lane = (q = lane & 15, keys 4 (lane >> 4) + r of the tile) ----
			const f16x8 qb0 = *(const f16x8*)( &L.qh[ qrow ][ lg * 8 ] ), qb1 = *(const f16x8*)( &L.qh[ qrow ][ 32 + lg * 8 ] );
			f32x4 sc[ M_TPW ];
			float mx = -INFINITY;
	#pragma unroll
			for( int j = 0; j < M_TPW; j++ )
			{
				const int t = wave + NW * j;
				sc[ j ] = f32x4{ -INFINITY, -INFINITY, -INFINITY, -INFINITY };
				if( t < nTiles )
				{
					f32x4 s4 = { 0.0f, 0.0f, 0.0f, 0.0f };
					s4 = __builtin_amdgcn_mfma_f32_16x16x32_f16( kf[ j ][ 0 ], qb0, s4, 0, 0, 0 );
					s4 = __builtin_amdgcn_mfma_f32_16x16x32_f16( kf[ j ][ 1 ], qb1, s4, 0, 0, 0 );
	#pragma unroll
					for( int r = 0; r < 4; r++ )
						if( t * 16 + lg * 4 + r < nk )
						{
							sc[ j ][ r ] = s4[ r ];
							mx = fmaxf( mx, s4[ r ] );
						}
				}
			}
	#pragma unroll
			for( int j = M_TPW / 2; j < M_TPW; j++ )
				if( wave + NW * ( j & ~1 ) < nTiles ) loadV( j );
			mx = fmaxf( mx, __shfl_xor( mx, 16, 64 ) );
			mx = fmaxf( mx, __shfl_xor( mx, 32, 64 ) );
			if( lg == 0 ) L.wmax[ wave ][ lr ] = mx;
			__syncthreads();
			mx = L.wmax[ 0 ][ lr ];
	#pragma unroll
			for( int w = 1; w < NW; w++ ) mx = fmaxf( mx, L.wmax[ w ][ lr ] );

			// ---- e = exp16( s - max ) (an FP16 value), double sum ----
			f16x4 pe[ M_TPW ];
			double sum = 0.0;
	#pragma unroll
			for( int j = 0; j < M_TPW; j++ )
	#pragma unroll
				for( int r = 0; r < 4; r++ )
				{
					const float e = sc[ j ][ r ] == -INFINITY ? 0.0f : exp16( sc[ j ][ r ] - mx );
					pe[ j ][ r ] = (f16)e;
					sum += (double)e;
				}
			sum += __shfl_xor( sum, 16, 64 );
			sum += __shfl_xor( sum, 32, 64 );
			if( lg == 0 ) L.wsum[ wave ][ lr ] = sum;

			// ---- O^T[dim][q] += V^T . P^T per block of 32 keys (two of the wave's tiles) ----
			f32x4 oa[ 4 ];
	#pragma unroll
			for( int dt = 0; dt < 4; dt++ ) oa[ dt ] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
			unsigned char* const vs = &L.vst[ wave ][ 0 ][ 0 ];
	#pragma unroll
			for( int jb = 0; jb < M_TPW / 2; jb++ )
			{
				if( wave + NW * ( 2 * jb ) >= nTiles ) continue;
	#pragma unroll
				for( int tt = 0; tt < 2; tt++ )
	#pragma unroll
					for( int i = 0; i < 2; i++ )
					{
						const int row = ( lane >> 3 ) + 8 * i;
						*(f16x8*)( vs + tt * 2048 + row * 128 + ( ( ( lane & 7 ) ^ ( ( ( row >> 2 ) & 3 ) << 1 ) ) << 4 ) ) = vf[ 2 * jb + tt ][ i ];
					}
				f16x8 pb;
	#pragma unroll
				for( int e = 0; e < 4; e++ )
				{
					pb[ e ] = pe[ 2 * jb ][ e ];
					pb[ 4 + e ] = pe[ 2 * jb + 1 ][ e ];
				}
	#pragma unroll
				for( int dt = 0; dt < 4; dt++ )
				{
					const int dim = dt * 16 + lr;
					f16x8 va;
	#pragma unroll
					for( int e = 0; e < 8; e++ )
						va[ e ] = *(const f16*)( vs + ( e >> 2 ) * 2048 + ( lg * 4 + ( e & 3 ) ) * 128 + ( ( ( dim >> 3 ) ^ ( lg << 1 ) ) << 4 ) + ( dim & 7 ) * 2 );
					oa[ dt ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( va, pb, oa[ dt ], 0, 0, 0 );
				}
			}
	#pragma unroll
			for( int dt = 0; dt < 4; dt++ )
	#pragma unroll
				for( int r = 0; r < 4; r++ ) L.part[ wave ][ dt ][ r ][ lane ] = oa[ dt ][ r ];
			__syncthreads();
			if( tid < NQ * HEAD_DIM )
			{
				const int q = tid >> 6, j = tid & 63;
				const int dt = j >> 4, ln = ( ( j & 15 ) >> 2 ) * 16 + q, r = j & 3;
				float t = L.part[ 0 ][ dt ][ r ][ ln ];
				double tot = L.wsum[ 0 ][ q ];
	#pragma unroll
				for( int w = 1; w < NW; w++ )
				{
					t += L.part[ w ][ dt ][ r ][ ln ];
					tot += L.wsum[ w ][ q ];
				}
				a.out[ ( (long long)bw * NQ + q ) * d + h * HEAD_DIM + j ] = (f16)( t * (float)( 1.0 / tot ) );
			}
		}

		template<int NQ>
		int launchDecM( const DecAttnArgs& a, hipStream_t stream )
		{
			constexpr int lds = (int)sizeof( DecMLds<NQ> );
			static_assert( lds <= 160 * 1024, "attentionDecM LDS" );
			static PerDeviceOnce once;
			if( const int onceDev = once.needed(); onceDev >= 0 )
			{
				WH_HIP( hipFuncSetAttribute( (const void*)attentionDecM<NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds ) );
				once.mark( onceDev );
			}
			hipLaunchKernelGGL( ( attentionDecM<NQ> ), dim3( a.H, a.batch / NQ ), dim3( NT ), lds, stream, a );
			WH_HIP( hipGetLastError() );
			return 0;
		}

		template<int NQ, bool FUSEQ, bool NT_LOADS = false>
		int launchDecG( const DecAttnArgs& a, hipStream_t stream )
		{
			constexpr int lds = (int)sizeof( DecGLds<NQ> );
			if( lds > 64 * 1024 )
			{
				static PerDeviceOnce once;
				if( const int onceDev = once.needed(); onceDev >= 0 )
				{
					WH_HIP( hipFuncSetAttribute( (const void*)attentionDecG<NQ, FUSEQ, NT_LOADS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds ) );
					once.mark( onceDev );
				}
			}
			hipLaunchKernelGGL( ( attentionDecG<NQ, FUSEQ, NT_LOADS> ), dim3( a.H, a.batch / NQ, a.nTok ), dim3( NT ), lds, stream, a );
			WH_HIP( hipGetLastError() );
			return 0;
		}
	}	// namespace

	int launchAttentionDec( const DecAttnArgs& a, hipStream_t stream )
	{
		if( a.nKeys <= 0 || a.nKeys > MAX_KEYS || a.nTok <= 0 || a.batch <= 0 )
		{
			setError( "attentionDec: key count out of range" );
			return -1;
		}
		const int group = a.group > 0 ? a.group : 1;
		const bool fuse = a.lnX != nullptr;
		if( ( a.batch % group ) != 0 || ( fuse && ( a.H * HEAD_DIM > G_MAXD || ( a.H * HEAD_DIM ) % 64 != 0 ) ) || ( a.parityThreads > NW && ( group > 1 || fuse ) ) )
		{
			setError( "attentionDec: unsupported group / fused-query configuration" );
			return -1;
		}
		// single-token causal self-attention of a big lock-step batch: a wave per (sequence, head)
		if( a.causal && a.nTok == 1 && group == 1 && !fuse && a.parityThreads <= 0 && a.keyStride <= SAW_MAX_KEYS && a.batch > g_opt.selfWaveMinRows )
		{
			hipLaunchKernelGGL( selfAttnDecWave, dim3( ( a.batch * a.H + SAW_WAVES - 1 ) / SAW_WAVES ), dim3( SAW_WAVES * 64 ), 0, stream, a );
			WH_HIP( hipGetLastError() );
			return 0;
		}
		if( !( g_tuning & TUNE_ATTN_DEC_G ) && group == 1 && !fuse )
		{
			hipLaunchKernelGGL( attentionDec, dim3( a.H, a.batch, a.nTok ), dim3( NT ), 0, stream, a );
			WH_HIP( hipGetLastError() );
			return 0;
		}
		// more than 8 virtual threads of the parity emulation only exist in the first kernel
		if( a.parityThreads > NW )
		{
			hipLaunchKernelGGL( attentionDec, dim3( a.H, a.batch, a.nTok ), dim3( NT ), 0, stream, a );
			WH_HIP( hipGetLastError() );
			return 0;
		}
		// the decode step's cross-attention (a query per window, fused query projection, all of a window's keys): streamed rows
		if( group == 1 && fuse && !a.causal && ( g_tuning & TUNE_ATTN_DEC_NT ) ) return launchDecG<1, true, true>( a, stream );
		// hypothesis groups (the cross-attention of a beam step): the matrix-core kernel (option cross_mfma)
		if( group > 1 && fuse && !a.causal && a.nTok == 1 && a.parityThreads <= 0 && g_opt.crossMfma )
		{
			switch( group )
			{
			case 2: return launchDecM<2>( a, stream );
			case 3: return launchDecM<3>( a, stream );
			case 4: return launchDecM<4>( a, stream );
			case 5: return launchDecM<5>( a, stream );
			case 8: return launchDecM<8>( a, stream );
			}
		}
	#define WH_DECG( N ) case N: return fuse ? launchDecG<N, true>( a, stream ) : launchDecG<N, false>( a, stream );
		switch( group )
		{
			WH_DECG( 1 )
			WH_DECG( 2 )
			WH_DECG( 3 )
			WH_DECG( 4 )
			WH_DECG( 5 )
			WH_DECG( 8 )
		}
	#undef WH_DECG
		setError( "attentionDec: hypothesis groups of 1, 2, 3, 4, 5 or 8 rows are supported" );
		return -1;
	}

	int launchSelfBlockDec( const DecSelfArgs& a, hipStream_t stream )
	{
		const int d = a.H * HEAD_DIM;
		if( a.batch <= 0 || d > G_MAXD || ( d % 64 ) != 0 || a.keyStride > MAX_KEYS / 3 || a.keyStride <= 0 )
		{
			setError( "selfBlockDec: unsupported shape (d <= 1280, n_text_ctx <= 512)" );
			return -1;
		}
		// sequences per workgroup: enough to bring the grid down to about one workgroup per CU -- the 384 KB weight slice of
		// a head is then read from L2 once per sequence group instead of once per sequence
		const int wgs1 = a.H * a.batch;
		const bool mf = ( g_tuning & TUNE_SELF_MFMA ) != 0;
		// (8 sequences per workgroup at 1792 pairs: 34.3 vs 32.7 us per launch -- the kernel is a latency chain, not L2-bound; not kept)
		// more than 128 sequences (one lock-step batch of 224 .. 448 windows): 8 per workgroup keep the grid at what 4 per workgroup are for 112 .. 224
		// (option self_nq pins 1 / 2 / 4 / 8 for A/B runs)
		const int nq = g_opt.selfNq;
		if( mf && ( nq == 8 || ( nq == 0 && wgs1 > 3584 ) ) ) return launchSelfBlockK<8, true>( a, stream );
		if( nq == 1 ) return mf ? launchSelfBlockK<1, true>( a, stream ) : launchSelfBlockK<1, false>( a, stream );
		if( nq == 2 ) return mf ? launchSelfBlockK<2, true>( a, stream ) : launchSelfBlockK<2, false>( a, stream );
		if( wgs1 > 768 || nq == 4 ) return mf ? launchSelfBlockK<4, true>( a, stream ) : launchSelfBlockK<4, false>( a, stream );
		if( wgs1 > 320 ) return mf ? launchSelfBlockK<2, true>( a, stream ) : launchSelfBlockK<2, false>( a, stream );
		return mf ? launchSelfBlockK<1, true>( a, stream ) : launchSelfBlockK<1, false>( a, stream );
	}
}
