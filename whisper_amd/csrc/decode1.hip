// Decode steps of ONE stream: the reference's own scenario (iContext::runFull feeds one sequence, one token at a time,
// Whisper/Whisper/ContextImpl.cpp:597-673; the layer is WhisperContext.cpp:407-576).
//
// A token of one sequence touches 35 MB of weights and cached keys per layer and does almost no arithmetic: the step is a
// chain of dependent launches, each bound by ONE memory round trip. Two things decide its length on MI355X:
//   * every launch must cover the chip. The batch kernels give a head or 16 weight rows to a workgroup (16 .. 64
//     workgroups at one sequence: 16.7 us for the cross-attention, 5.2 us for a 2 MB product). Here every product is
//     ~256 workgroups of 4 waves with all of a wave's loads in flight at once, and the cross-attention is H x 8 workgroups
//     over disjoint key ranges, in two launches because the reference's table softmax needs the maximum over ALL keys
//     before the first exponential (ggml.c:5030-5090; a split with local maxima would round differently);
//   * the round trip should end in the L2 of the XCD that asks, not in HBM. The launches of a step are known in advance, so
//     each launch carries 256 extra workgroups (dispatched after the computing ones) that do nothing but touch the lines the
//     NEXT launch will stream -- the weight rows workgroup c of that launch reads, requested from a workgroup on XCD c % 8,
//     the XCD workgroup c will be dispatched to. HBM is idle during a decode step (0.7 TB/s average), the 4 MiB L2 of an XCD holds
//     its eighth of the largest matrix four times over. The guess about placement costs nothing when it is wrong.
// Arithmetic: products accumulate in FP32 (v_dot2_f32_f16 per pair of FP16 operands), LayerNorm and the epilogues are
// the ones of the batch kernels (epilogue.h, common.h), so results differ from gemvFused / attentionDecG by FP32 summation
// order only; tests/test_gpu_model.py compares the two paths step by step.
#include "kernels.h"
#include "epilogue.h"

namespace wh
{
	namespace
	{
		constexpr int GS_WAVES = 4;				  // waves of a gemvSmall workgroup
		constexpr int GS_NT = GS_WAVES * 64;
		constexpr int GS_SLOTS = 16;			  // 16-byte weight loads a lane has in flight
		constexpr int CS_NT = 512;				  // threads of the cross-attention kernels (8 lanes per K/V row, 64 rows per pass)
		constexpr int CS_WAVES = CS_NT / 64;
		constexpr int CS_MAXD = 1280;
		constexpr int CS_MAX_KEYS = 1536;
		constexpr int CS_PASSES = 3;			  // 64-row passes per split: 8 x 3 x 64 = 1536 keys
		constexpr int PF_WGS = 256;				  // prefetch workgroups appended to a grid (a multiple of 8: 32 per XCD)

		// ---- the prefetch workgroups ----------------------------------------------------------------------------------------
		// Consumer workgroup c of the NEXT launch streams bytes [c * chunk, (c + 1) * chunk) of the region and will run on XCD
		// c % 8. A prefetch workgroup with linear id wg runs on XCD wg % 8 and is the `me`-th of the PF_WGS / 8 prefetchers there:
		// it takes its share of every chunk that belongs to its XCD. 16 bytes per lane and instruction, every byte of a line is
		// requested (a narrower load may fill only a sector). The workgroup ends when its lines have arrived (s_endpgm waits
		// for outstanding loads), so the launch is not over before the next launch's weights are on the chip.
		__device__ __forceinline__ void prefetchRegion( const PrefetchHint& h, int xcd, int me, int per, int tid, int nThreads, unsigned ldsBase )
		{
			if( h.ptr == nullptr || h.bytes <= 0 || h.chunkBytes <= 0 ) return;
			const long long nChunks = ( h.bytes + h.chunkBytes - 1 ) / h.chunkBytes;
			const int pieces = ( h.chunkBytes + 15 ) >> 4;	  // 16-byte pieces of a chunk
			const int share = ( ( pieces + per - 1 ) / per + 63 ) & ~63;   // whole 1 KiB wave instructions
			const int first = me * share;
			const char* const base = (const char*)h.ptr;
			// (chunk of this XCD, piece of this prefetcher's share) pairs, flattened over the threads: share is a multiple of 64, so
			// the 64 lanes of an instruction always cover 1 KiB of one chunk
			const long long mine = ( nChunks - xcd + 7 ) / 8;	  // chunks c = xcd, xcd + 8, ...
			const long long total = mine * share;
			for( long long f = tid; f < total; f += nThreads )
			{
				const long long ci = f / share;
				const int p = first + (int)( f - ci * share );
				const long long off = ( xcd + 8 * ci ) * h.chunkBytes + (long long)p * 16;
				if( p < pieces && off + 16 <= h.bytes )
				{
					// global -> LDS: the data has no register destination (a plain load still in flight would land in a register the
					// compiler has handed to something else by then); it lands in the first KiB of this workgroup's LDS, which a
					// prefetch workgroup never reads. M0 (the LDS base) is saved and restored around the statement.
					unsigned keep;
					asm volatile( "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
								  : "=&s"( keep )
								  : "v"( base + off ), "s"( ldsBase )
								  : "memory" );
				}
			}
		}
		// lds = at least 1 KiB of this workgroup's LDS
		__device__ __forceinline__ void prefetchWorkgroup( const PrefetchHint ( &pf )[ 2 ], int wg, int tid, int nThreads, void* lds )
		{
			typedef __attribute__( ( address_space( 3 ) ) ) void* LdsPtr;
			const unsigned ldsBase = __builtin_amdgcn_readfirstlane( (unsigned)(size_t)(LdsPtr)lds );
			const int xcd = wg & 7;
			const int me = ( wg >> 3 ) % ( PF_WGS / 8 );
			prefetchRegion( pf[ 0 ], xcd, me, PF_WGS / 8, tid, nThreads, ldsBase );
			prefetchRegion( pf[ 1 ], xcd, me, PF_WGS / 8, tid, nThreads, ldsBase );
			asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
		}

		__device__ __forceinline__ float dot8( const f16x8 w, const f16x8 x, float acc )
		{
#pragma unroll
			for( int e = 0; e < 4; e++ )
			{
				const f16x2 a = { w[ 2 * e ], w[ 2 * e + 1 ] };
				const f16x2 b = { x[ 2 * e ], x[ 2 * e + 1 ] };
				acc = __builtin_amdgcn_fdot2( a, b, acc, false );
			}
			return acc;
		}

		// LayerNorm of MR rows of length d (<= 2048) by GS_WAVES waves, FP16 result into LDS: the arithmetic of layerNormRows
		// (common.h; two passes in FP32, every rounding spelled out), the reduction tree spans the workgroup.
		template<int MR>
		__device__ __forceinline__ void layerNormToLds( const float* __restrict__ x, int nRows, const float* __restrict__ w, const float* __restrict__ b, int d,
			int tid, f16* xs, int xsStride, float ( &red )[ 2 ][ GS_WAVES ][ MR ] )
		{
			constexpr int NTH = GS_WAVES * 64;
			const int nv = d >> 2;
			const int lane = tid & 63, wave = tid >> 6;
			f32x4 v[ MR ][ 2 ], wv[ 2 ], bv[ 2 ];
			bool own[ 2 ];
#pragma unroll
			for( int i = 0; i < 2; i++ )
			{
				const int c = tid + i * NTH;
				own[ i ] = c < nv;
				const int cc = own[ i ] ? c : 0;
				wv[ i ] = *(const f32x4*)( w + cc * 4 );
				bv[ i ] = *(const f32x4*)( b + cc * 4 );
#pragma unroll
				for( int m = 0; m < MR; m++ )
				{
					const int mr = m < nRows ? m : nRows - 1;
					v[ m ][ i ] = *(const f32x4*)( x + (long long)mr * d + cc * 4 );
				}
			}
			const float invD = 1.0f / (float)d;
			float s[ MR ];
#pragma unroll
			for( int m = 0; m < MR; m++ )
			{
				s[ m ] = 0.0f;
#pragma unroll
				for( int i = 0; i < 2; i++ )
					if( own[ i ] ) s[ m ] = __fadd_rn( s[ m ], __fadd_rn( __fadd_rn( v[ m ][ i ][ 0 ], v[ m ][ i ][ 1 ] ), __fadd_rn( v[ m ][ i ][ 2 ], v[ m ][ i ][ 3 ] ) ) );
				s[ m ] = waveReduceSum( s[ m ] );
			}
			if( lane == 0 )
#pragma unroll
				for( int m = 0; m < MR; m++ ) red[ 0 ][ wave ][ m ] = s[ m ];
			__syncthreads();
			float q[ MR ];
#pragma unroll
			for( int m = 0; m < MR; m++ )
			{
				float t = red[ 0 ][ 0 ][ m ];
#pragma unroll
				for( int ww = 1; ww < GS_WAVES; ww++ ) t = __fadd_rn( t, red[ 0 ][ ww ][ m ] );
				const float mean = __fmul_rn( t, invD );
				q[ m ] = 0.0f;
#pragma unroll
				for( int i = 0; i < 2; i++ )
				{
#pragma unroll
					for( int e = 0; e < 4; e++ ) v[ m ][ i ][ e ] = __fsub_rn( v[ m ][ i ][ e ], mean );
					if( own[ i ] )
					{
						float t2 = __fmul_rn( v[ m ][ i ][ 0 ], v[ m ][ i ][ 0 ] );
						t2 = fmaf( v[ m ][ i ][ 1 ], v[ m ][ i ][ 1 ], t2 );
						t2 = fmaf( v[ m ][ i ][ 2 ], v[ m ][ i ][ 2 ], t2 );
						t2 = fmaf( v[ m ][ i ][ 3 ], v[ m ][ i ][ 3 ], t2 );
						q[ m ] = __fadd_rn( q[ m ], t2 );
					}
				}
				q[ m ] = waveReduceSum( q[ m ] );
			}
			if( lane == 0 )
#pragma unroll
				for( int m = 0; m < MR; m++ ) red[ 1 ][ wave ][ m ] = q[ m ];
			__syncthreads();
#pragma unroll
			for( int m = 0; m < MR; m++ )
			{
				float t = red[ 1 ][ 0 ][ m ];
#pragma unroll
				for( int ww = 1; ww < GS_WAVES; ww++ ) t = __fadd_rn( t, red[ 1 ][ ww ][ m ] );
				const float scale = 1.0f / sqrtf( __fadd_rn( __fmul_rn( t, invD ), 1e-5f ) );
#pragma unroll
				for( int i = 0; i < 2; i++ )
					if( own[ i ] )
					{
						f16x4 hv;
#pragma unroll
						for( int e = 0; e < 4; e++ ) hv[ e ] = (f16)__fadd_rn( __fmul_rn( __fmul_rn( v[ m ][ i ][ e ], scale ), wv[ i ][ e ] ), bv[ i ][ e ] );
						*(f16x4*)( xs + m * xsStride + ( tid + i * NTH ) * 4 ) = hv;
					}
			}
		}

		// ---------------------------------------------------------------------------------------------------------------
		// gemvSmall: out[m][n] = epilogue( sum_k W[n][k] * x[m][k] ) for MR <= 4 activation rows.
		// grid = ceil( N / rowsPerWg ) workgroups (the launcher aims at ~256), a wave owns rowsPerWg / 4 consecutive weight rows =
		// one contiguous span of memory, read as 1 KiB pieces (16 bytes per lane), ALL of them requested before anything else
		// (at most GS_SLOTS per lane; the launcher sizes rowsPerWg accordingly). The activations are built once per workgroup
		// in LDS by the prologue. A row's 64 partial sums meet in a shuffle tree; the epilogue of the workgroup's rows runs
		// on its first threads.
		// ---------------------------------------------------------------------------------------------------------------
		template<int EPI, int PRO, int MR>
		__global__ void __launch_bounds__( GS_NT ) gemvSmall( const SmallGemvArgs a, const int rowsPerWg, const int nWgCompute )
		{
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char gsLds[];
			__shared__ float red[ 2 ][ GS_WAVES ][ MR ];
			__shared__ float results[ 64 * MR ];

			const GemmArgs& g = a.g;
			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
			if( (int)blockIdx.x >= nWgCompute )
			{
				prefetchWorkgroup( a.pf, blockIdx.x, tid, GS_NT, gsLds );
				return;
			}
			const int K = g.K;
			const int nch = ( K + 511 ) >> 9;		// 1 KiB pieces per weight row
			const int kPad = nch << 9;
			f16* const xs = (f16*)gsLds;			// [MR][kPad]
			const int rw = rowsPerWg / GS_WAVES;
			const int nSlots = rw * nch;
			const int n0 = blockIdx.x * rowsPerWg + wave * rw;

			// ---- all weight loads of the wave ----
			f16x8 w[ GS_SLOTS ];
			{
				int r = 0, j = 0;
#pragma unroll
				for( int s = 0; s < GS_SLOTS; s++ )
				{
					if( s < nSlots )
					{
						int n = n0 + r;
						n = n < g.N ? n : g.N - 1;
						int k = ( j * 64 + lane ) * 8;
						k = k < K ? k : 0;
						w[ s ] = __builtin_nontemporal_load( (const f16x8*)( g.W + (long long)n * K + k ) );
						if( ++j == nch )
						{
							j = 0;
							r++;
						}
					}
				}
			}

			// ---- activations into LDS ----
			if constexpr( PRO == 1 )
			{
				layerNormToLds<MR>( g.lnX, g.M, g.lnW, g.lnB, K, tid, xs, kPad, red );
				// columns K .. kPad - 1 (K not a multiple of 512) are multiplied with a clamped weight load: zero them
				for( int c = K + tid; c < kPad; c += GS_WAVES * 64 )
#pragma unroll
					for( int m = 0; m < MR; m++ ) xs[ m * kPad + c ] = (f16)0.0f;
			}
			else if constexpr( PRO == 3 )
			{
				// x[m][h * 64 + j] = fp16( ( sum over the splits, in order, of part[m][h][s][j] ) * float( 1 / sum over the splits of the double sums ) )
				const int H = K >> 6;
				for( int c4 = tid; c4 < ( kPad >> 2 ); c4 += GS_WAVES * 64 )
				{
					const int col = c4 * 4;
					const int h = col >> 6, j = col & 63;
#pragma unroll
					for( int m = 0; m < MR; m++ )
					{
						f16x4 hv = { (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f };
						if( col < K && m < g.M )
						{
							const float* const p = a.part + ( (long long)m * H + h ) * CROSS_SPLITS * CROSS_PART;
							f32x4 acc = *(const f32x4*)( p + j );
							double tot = *(const double*)( p + 64 );
#pragma unroll
							for( int s = 1; s < CROSS_SPLITS; s++ )
							{
								const f32x4 t = *(const f32x4*)( p + s * CROSS_PART + j );
#pragma unroll
								for( int e = 0; e < 4; e++ ) acc[ e ] += t[ e ];
								tot += *(const double*)( p + s * CROSS_PART + 64 );
							}
							const float inv = (float)( 1.0 / tot );
#pragma unroll
							for( int e = 0; e < 4; e++ ) hv[ e ] = (f16)( acc[ e ] * inv );
						}
						*(f16x4*)( xs + m * kPad + col ) = hv;
					}
				}
			}
			else
			{
				for( int c8 = tid; c8 < ( kPad >> 3 ); c8 += GS_WAVES * 64 )
				{
					const int col = c8 * 8;
#pragma unroll
					for( int m = 0; m < MR; m++ )
					{
						f16x8 hv = { (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f };
						if( col < K && m < g.M ) hv = *(const f16x8*)( g.A + rowOffset( m, g.Mb, g.lda, g.aBatchStride ) + col );
						*(f16x8*)( xs + m * kPad + col ) = hv;
					}
				}
			}
			__syncthreads();

			// ---- products ----
			{
				float acc[ MR ];
#pragma unroll
				for( int m = 0; m < MR; m++ ) acc[ m ] = 0.0f;
				int r = 0, j = 0;
#pragma unroll
				for( int s = 0; s < GS_SLOTS; s++ )
				{
					if( s < nSlots )
					{
#pragma unroll
						for( int m = 0; m < MR; m++ )
						{
							const f16x8 xv = *(const f16x8*)( xs + m * kPad + ( j * 64 + lane ) * 8 );
							acc[ m ] = dot8( w[ s ], xv, acc[ m ] );
						}
						if( ++j == nch )
						{
#pragma unroll
							for( int m = 0; m < MR; m++ )
							{
								const float t = waveReduceSum( acc[ m ] );
								if( lane == 0 ) results[ ( wave * rw + r ) * MR + m ] = t;
								acc[ m ] = 0.0f;
							}
							j = 0;
							r++;
						}
					}
				}
			}
			__syncthreads();
			if( tid < rowsPerWg * MR )
			{
				const int row = tid / MR, m = tid - row * MR;
				const int n = blockIdx.x * rowsPerWg + row;
				if( n < g.N && m < g.M ) epilogueOne<EPI>( g, m, n, results[ tid ] );
			}
		}

		// ---------------------------------------------------------------------------------------------------------------
		// Cross-attention, first launch: grid (H, CROSS_SPLITS, sequences). K rows of the split are requested first; then the
		// LayerNorm of the residual row and this head's 64 query rows (8 lanes per weight row, the arithmetic of attentionDecG's
		// fused query, so the query -- and with it every score -- has the same bits there and here); then K . q.
		// ---------------------------------------------------------------------------------------------------------------
		struct CrossScoresLds
		{
			float qs[ HEAD_DIM ];
			float shf[ CS_WAVES ];
			f16 xn[ CS_MAXD ];
		};

		__device__ __forceinline__ float xorReduce8( float v )
		{
			v += __shfl_xor( v, 1, 64 );
			v += __shfl_xor( v, 2, 64 );
			v += __shfl_xor( v, 4, 64 );
			return v;
		}

		__global__ void __launch_bounds__( CS_NT ) crossScores( const CrossSplitArgs a )
		{
			__shared__ CrossScoresLds L;
			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
			const int nWgCompute = a.H * CROSS_SPLITS * a.batch;
			if( (int)blockIdx.x >= nWgCompute )
			{
				prefetchWorkgroup( a.pf, blockIdx.x, tid, CS_NT, &L );
				return;
			}
			const int g = tid >> 3, c = tid & 7;
			// head fastest: the splits of a head run on one XCD (H % 8 == 0), which then holds that head's query rows and K / V once
			const int h = blockIdx.x % a.H, sp = ( blockIdx.x / a.H ) % CROSS_SPLITS, b = blockIdx.x / ( a.H * CROSS_SPLITS );
			const int d = a.H * HEAD_DIM;
			const int per = ( ( a.nKeys + CROSS_SPLITS - 1 ) / CROSS_SPLITS + 3 ) & ~3;
			const int key0 = sp * per;
			const int keyEnd = min( key0 + per, a.nKeys );
			const f16* const K = a.kc + ( (long long)b * a.H + h ) * a.keyStride * HEAD_DIM;

			f16x8 kv[ CS_PASSES ];
#pragma unroll
			for( int u = 0; u < CS_PASSES; u++ )
			{
				int key = key0 + u * 64 + g;
				key = key < a.nKeys ? key : a.nKeys - 1;
				kv[ u ] = *(const f16x8*)( K + (long long)key * HEAD_DIM + c * 8 );
			}

			// the head's query weight rows do not depend on the LayerNorm either: all of this thread's fragments (d / 64 <= 20 loads
			// of 16 bytes, 8 lanes per weight row) are requested now, one memory round trip for K, the residual row and the weights
			constexpr int WQ_MAX = CS_MAXD / 64;
			const int steps = d >> 6;
			f16x8 wq[ WQ_MAX ];
			{
				const f16* const wr = a.qW + ( (long long)h * HEAD_DIM + g ) * d + c * 8;
#pragma unroll
				for( int u = 0; u < WQ_MAX; u++ )
					if( u < steps ) wq[ u ] = *(const f16x8*)( wr + u * 64 );
			}

			// ---- LayerNorm of the residual row (FP32 two-pass like attentionDecG) ----
			{
				const int nv = d >> 2;
				const bool own = tid < nv;
				f32x4 xv = { 0, 0, 0, 0 }, wv = { 0, 0, 0, 0 }, bv = { 0, 0, 0, 0 };
				if( own )
				{
					wv = *(const f32x4*)( a.lnW + tid * 4 );
					bv = *(const f32x4*)( a.lnB + tid * 4 );
					xv = *(const f32x4*)( a.lnX + (long long)b * d + tid * 4 );
				}
				const float invD = 1.0f / (float)d;
				float s = waveReduceSum( ( xv[ 0 ] + xv[ 1 ] ) + ( xv[ 2 ] + xv[ 3 ] ) );
				if( lane == 0 ) L.shf[ wave ] = s;
				__syncthreads();
				float t = L.shf[ 0 ];
#pragma unroll
				for( int w = 1; w < CS_WAVES; w++ ) t += L.shf[ w ];
				const float mean = t * invD;
				__syncthreads();
				float q2 = 0.0f;
				if( own )
				{
#pragma unroll
					for( int e = 0; e < 4; e++ )
					{
						xv[ e ] -= mean;
						q2 = fmaf( xv[ e ], xv[ e ], q2 );
					}
				}
				s = waveReduceSum( q2 );
				if( lane == 0 ) L.shf[ wave ] = s;
				__syncthreads();
				t = L.shf[ 0 ];
#pragma unroll
				for( int w = 1; w < CS_WAVES; w++ ) t += L.shf[ w ];
				const float rstd = 1.0f / sqrtf( t * invD + 1e-5f );
				if( own )
				{
					f16x4 hv;
#pragma unroll
					for( int e = 0; e < 4; e++ ) hv[ e ] = (f16)__fadd_rn( __fmul_rn( __fmul_rn( xv[ e ], rstd ), wv[ e ] ), bv[ e ] );
					*(f16x4*)( &L.xn[ tid * 4 ] ) = hv;
				}
				__syncthreads();
			}
			// ---- q[j] = fp16( ( W[h*64 + j] . xn + bias ) * scale ): 8 lanes per weight row, same order of the products as attentionDecG ----
			{
				float acc = 0.0f;
#pragma unroll
				for( int u = 0; u < WQ_MAX; u++ )
					if( u < steps )
					{
						const f16x8 xq = *(const f16x8*)( &L.xn[ u * 64 + c * 8 ] );
#pragma unroll
						for( int e = 0; e < 8; e++ ) acc = fmaf( (float)wq[ u ][ e ], (float)xq[ e ], acc );
					}
				const float t = xorReduce8( acc );
				if( c == 0 ) L.qs[ g ] = round16( ( t + a.qB[ h * HEAD_DIM + g ] ) * a.qScale );
				__syncthreads();
			}
			float qf[ 8 ];
#pragma unroll
			for( int e = 0; e < 8; e++ ) qf[ e ] = L.qs[ c * 8 + e ];

			// ---- scores of the split ----
			float mx = -INFINITY;
			float* const sc = a.scores + ( (long long)b * a.H + h ) * a.keyStride;
#pragma unroll
			for( int u = 0; u < CS_PASSES; u++ )
			{
				const int key = key0 + u * 64 + g;
				float sacc = 0.0f;
#pragma unroll
				for( int e = 0; e < 8; e++ ) sacc = fmaf( (float)kv[ u ][ e ], qf[ e ], sacc );
				sacc = xorReduce8( sacc );
				if( key < keyEnd )
				{
					mx = fmaxf( mx, sacc );
					if( c == 0 ) sc[ key ] = sacc;
				}
			}
			mx = waveReduceMax( mx );
			if( lane == 0 ) L.shf[ wave ] = mx;
			__syncthreads();
			if( tid == 0 )
			{
				float m = L.shf[ 0 ];
#pragma unroll
				for( int w = 1; w < CS_WAVES; w++ ) m = fmaxf( m, L.shf[ w ] );
				a.splitMax[ ( (long long)b * a.H + h ) * CROSS_SPLITS + sp ] = m;
			}
		}

		// ---------------------------------------------------------------------------------------------------------------
		// Cross-attention, second launch: same grid. V rows of the split first; the maximum over the splits; e = exp16( s - max )
		// (the reference's table, common.h), its sum in double, sum( e * V ) in FP32 per dimension.
		// ---------------------------------------------------------------------------------------------------------------
		__global__ void __launch_bounds__( CS_NT ) crossSoftmaxPV( const CrossSplitArgs a )
		{
			__shared__ float red[ CS_WAVES ][ HEAD_DIM ];
			__shared__ double shd[ CS_WAVES ];
			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
			const int nWgCompute = a.H * CROSS_SPLITS * a.batch;
			if( (int)blockIdx.x >= nWgCompute )
			{
				prefetchWorkgroup( a.pf, blockIdx.x, tid, CS_NT, red );
				return;
			}
			const int g = tid >> 3, c = tid & 7;
			// head fastest: the splits of a head run on one XCD (H % 8 == 0), which then holds that head's query rows and K / V once
			const int h = blockIdx.x % a.H, sp = ( blockIdx.x / a.H ) % CROSS_SPLITS, b = blockIdx.x / ( a.H * CROSS_SPLITS );
			const int per = ( ( a.nKeys + CROSS_SPLITS - 1 ) / CROSS_SPLITS + 3 ) & ~3;
			const int key0 = sp * per;
			const int keyEnd = min( key0 + per, a.nKeys );
			const long long bh = (long long)b * a.H + h;
			const f16* const V = a.vc + bh * a.keyStride * HEAD_DIM;
			const float* const sc = a.scores + bh * a.keyStride;

			f16x8 vv[ CS_PASSES ];
			float sv[ CS_PASSES ];
#pragma unroll
			for( int u = 0; u < CS_PASSES; u++ )
			{
				int key = key0 + u * 64 + g;
				key = key < a.nKeys ? key : a.nKeys - 1;
				vv[ u ] = *(const f16x8*)( V + (long long)key * HEAD_DIM + c * 8 );
				sv[ u ] = sc[ key ];
			}
			float mx = a.splitMax[ bh * CROSS_SPLITS ];
#pragma unroll
			for( int s = 1; s < CROSS_SPLITS; s++ ) mx = fmaxf( mx, a.splitMax[ bh * CROSS_SPLITS + s ] );

			float acc[ 8 ];
#pragma unroll
			for( int e = 0; e < 8; e++ ) acc[ e ] = 0.0f;
			double sum = 0.0;
#pragma unroll
			for( int u = 0; u < CS_PASSES; u++ )
			{
				const int key = key0 + u * 64 + g;
				const float ev = key < keyEnd ? exp16( sv[ u ] - mx ) : 0.0f;
				if( c == 0 ) sum += (double)ev;
#pragma unroll
				for( int e = 0; e < 8; e++ ) acc[ e ] = fmaf( (float)vv[ u ][ e ], ev, acc[ e ] );
			}
			// the 8 row slots of a wave (lanes with equal c), then the waves through LDS in a fixed order
#pragma unroll
			for( int e = 0; e < 8; e++ )
			{
				float t = acc[ e ];
				t += __shfl_xor( t, 8, 64 );
				t += __shfl_xor( t, 16, 64 );
				t += __shfl_xor( t, 32, 64 );
				acc[ e ] = t;
			}
			sum = waveReduceSumD( sum );
			if( lane < 8 )
#pragma unroll
				for( int e = 0; e < 8; e++ ) red[ wave ][ lane * 8 + e ] = acc[ e ];
			if( lane == 0 ) shd[ wave ] = sum;
			__syncthreads();
			float* const out = a.part + ( bh * CROSS_SPLITS + sp ) * CROSS_PART;
			if( tid < HEAD_DIM )
			{
				float t = red[ 0 ][ tid ];
#pragma unroll
				for( int w = 1; w < CS_WAVES; w++ ) t += red[ w ][ tid ];
				out[ tid ] = t;
			}
			if( tid == HEAD_DIM )
			{
				double t = shd[ 0 ];
#pragma unroll
				for( int w = 1; w < CS_WAVES; w++ ) t += shd[ w ];
				*(double*)( out + 64 ) = t;
			}
		}

		template<int EPI, int PRO, int MR>
		int launchGemvSmallK( const SmallGemvArgs& a, int rowsPerWg, hipStream_t stream )
		{
			const int kPad = ( ( a.g.K + 511 ) >> 9 ) << 9;
			const size_t lds = (size_t)MR * kPad * 2;
			if( lds > 48 * 1024 )
			{
				static PerDeviceOnce once;
				if( const int onceDev = once.needed(); onceDev >= 0 )
				{
					WH_HIP( hipFuncSetAttribute( (const void*)gemvSmall<EPI, PRO, MR>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 ) );
					once.mark( onceDev );
				}
			}
			const int grid = ( a.g.N + rowsPerWg - 1 ) / rowsPerWg;
			const int extra = ( a.pf[ 0 ].ptr || a.pf[ 1 ].ptr ) ? PF_WGS : 0;
			hipLaunchKernelGGL( ( gemvSmall<EPI, PRO, MR> ), dim3( grid + extra ), dim3( GS_NT ), lds, stream, a, rowsPerWg, grid );
			WH_HIP( hipGetLastError() );
			return 0;
		}

		template<int EPI, int PRO>
		int launchGemvSmallM( const SmallGemvArgs& a, int rowsPerWg, hipStream_t stream )
		{
			if( a.g.M == 1 ) return launchGemvSmallK<EPI, PRO, 1>( a, rowsPerWg, stream );
			if( a.g.M == 2 ) return launchGemvSmallK<EPI, PRO, 2>( a, rowsPerWg, stream );
			return launchGemvSmallK<EPI, PRO, 4>( a, rowsPerWg, stream );
		}
	}	// namespace

	int launchGemvSmall( const SmallGemvArgs& a, hipStream_t stream )
	{
		const GemmArgs& g = a.g;
		const int nch = ( g.K + 511 ) >> 9;
		if( g.M <= 0 || g.M > SMALL_MAX_ROWS || g.N <= 0 || g.K <= 0 || ( g.K % 8 ) != 0 || nch > GS_SLOTS || ( a.pro == 1 && g.K > 2048 ) ||
			( a.pro == 3 && ( g.K % 64 ) != 0 ) )
		{
			setError( "gemvSmall: need 0 < M <= 4, K a multiple of 8 and at most 8192 (2048 with the LayerNorm prologue)" );
			return -1;
		}
		// rows per wave: the chip is covered by ~256 workgroups, a lane keeps at most GS_SLOTS loads in flight, a workgroup's
		// results fit the 64-entry epilogue stage
		int rw = ( g.N + 256 * GS_WAVES - 1 ) / ( 256 * GS_WAVES );
		if( rw < 1 ) rw = 1;
		const int mr = g.M <= 2 ? g.M : 4;	  // the instantiated row count
		while( rw > 1 && ( rw * nch > GS_SLOTS || rw * GS_WAVES * mr > 64 ) ) rw--;
		const int rowsPerWg = rw * GS_WAVES;
#define WH_GS( E )                                                                      \
	case E:                                                                             \
		if( a.pro == 1 ) return launchGemvSmallM<E, 1>( a, rowsPerWg, stream );          \
		if( a.pro == 3 ) return launchGemvSmallM<E, 3>( a, rowsPerWg, stream );          \
		return launchGemvSmallM<E, 0>( a, rowsPerWg, stream );
		switch( g.epi )
		{
			WH_GS( EPI_F32 )
			WH_GS( EPI_F16_GELU )
			WH_GS( EPI_QKV_DEC )
		}
#undef WH_GS
		setError( "gemvSmall: epilogue not available" );
		return -1;
	}

	static int checkCross( const CrossSplitArgs& a )
	{
		const int d = a.H * HEAD_DIM;
		if( a.batch <= 0 || a.batch > SMALL_MAX_ROWS || a.nKeys <= 0 || a.nKeys > CS_MAX_KEYS || a.nKeys > a.keyStride || d > CS_MAXD || ( d % 64 ) != 0 ||
			( ( ( a.nKeys + CROSS_SPLITS - 1 ) / CROSS_SPLITS + 3 ) & ~3 ) > CS_PASSES * 64 )
		{
			setError( "crossSplit: unsupported shape (up to 4 sequences, 1536 keys, d <= 1280)" );
			return -1;
		}
		return 0;
	}

	int launchCrossScores( const CrossSplitArgs& a, hipStream_t stream )
	{
		WH_CHECK( checkCross( a ) );
		const int extra = ( a.pf[ 0 ].ptr || a.pf[ 1 ].ptr ) ? PF_WGS : 0;
		hipLaunchKernelGGL( crossScores, dim3( a.H * CROSS_SPLITS * a.batch + extra ), dim3( CS_NT ), 0, stream, a );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchCrossSoftmaxPV( const CrossSplitArgs& a, hipStream_t stream )
	{
		WH_CHECK( checkCross( a ) );
		const int extra = ( a.pf[ 0 ].ptr || a.pf[ 1 ].ptr ) ? PF_WGS : 0;
		hipLaunchKernelGGL( crossSoftmaxPV, dim3( a.H * CROSS_SPLITS * a.batch + extra ), dim3( CS_NT ), 0, stream, a );
		WH_HIP( hipGetLastError() );
		return 0;
	}
}	// namespace wh
