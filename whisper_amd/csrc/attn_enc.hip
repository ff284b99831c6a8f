// Encoder self-attention for gfx950, exact ggml_flash_attn_f16 semantics with the scores kept in registers.
//
// Replaces the encoder's mulMat(K,Q) -> softMax(1/8) -> mulMat(V,.) chain (Whisper/ML/Context.ops.cpp:199-213) and
// flashAttention.hlsl:76-169 of the reference. Numerics follow the reference CPU path
// (Whisper/source/ggml.c:5912-6097): S = K.Q with FP16 operands and FP32 accumulation, * 1/sqrt(64); row maximum
// over ALL keys; e = exp16(S - max) (the FP16 table semantics); sum in double; P = fp16(e * float(1/sum)); O = V.P
// with FP32 accumulation. Because P must be rounded to FP16 AFTER normalising by the full row sum, an online
// (running-max) softmax cannot reproduce it; instead one workgroup owns 64 query rows x ALL keys:
//
//   * 512 threads = 8 waves; wave w owns keys [w*32*KT, (w+1)*32*KT) (KT = 6 for n_ctx = 1500 -> 1536 padded keys).
//   * S^T tiles (32 keys x 32 queries) come from v_mfma_f32_32x32x16_f16 with K as the A operand straight from
//     global/L2 (each K row is read by exactly one wave, so LDS staging would be pure overhead) and the Q tile as the
//     B operand from LDS. The accumulators (KT*2 tiles x 16 registers) never leave the register file.
//   * In the 32x32 accumulator layout a lane holds one query column and 16 keys per tile, so row max / row sum are
//     in-lane loops + one xor-32 shuffle + an 8-entry LDS exchange between waves.
//   * P is packed to FP16 in registers directly in the B-operand layout of the P.V MFMA: the accumulator's key order
//     (r&3) + 8*(r>>2) + 4*(lane>>5) is matched by permuting the K-slots of the V operand, which the QKV GEMM epilogue
//     stores fragment-major (gemm.hip vFragIndex) so that each operand is one coalesced 16-byte load per lane; no
//     cross-lane movement is needed.
//   * the 8 partial O^T tiles are tree-reduced through LDS in a fixed order (deterministic).
#include "kernels.h"
#include <type_traits>

namespace wh
{
	namespace
	{
		constexpr int AQ = 64;				   // query rows per workgroup
		constexpr int AWAVES = 8;
		constexpr int QSTRIDE = 72;			   // halfs per LDS row of the Q tile
		constexpr int OSTRIDE = 65;			   // floats per LDS row of an O buffer
		constexpr int ATT_LDS_BYTES = AQ * QSTRIDE * 2 + 2 * AWAVES * AQ * 4 + AWAVES * AQ * 8 + 4 * AQ * OSTRIDE * 4;

		template<int KT>
		__global__ void __launch_bounds__( 512, 2 ) attentionEnc( const f16* __restrict__ q, const f16* __restrict__ k,
			const f16* __restrict__ vT, f16* __restrict__ out, int heads, int T, int Tpad, int nQ, int xcdRemap )
		{
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smem[];
			f16* const ldsQ = (f16*)smem;
			float* const redMax = (float*)( smem + AQ * QSTRIDE * 2 );
			float* const redInv = redMax + AWAVES * AQ;
			double* const redSum = (double*)( redInv + AWAVES * AQ );
			float* const obuf = (float*)( redSum + AWAVES * AQ );

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = tid >> 6;
			const int hi = lane >> 5;
			const int c = lane & 31;
			// Workgroups are handed to the 8 XCDs round-robin by linear id. All query blocks of one (sequence, head) read the
			// same 384 KB of K and V, so they are mapped to ONE XCD, back to back: K/V then come from that XCD's L2 instead of
			// being fetched once per XCD (measured before the remap: 418 MB of fabric reads per launch for 64 MB of operands,
			// profiles/r01_pmc_hbm_traffic.csv). xcdRemap == 0 keeps the plain (query block, bh) order.
			int bh, qb;
			{
				const int L = blockIdx.x;
				if( xcdRemap )
				{
					const int kIdx = L >> 3;
					bh = ( L & 7 ) + 8 * ( kIdx / nQ );
					qb = kIdx % nQ;
				}
				else
				{
					bh = L / nQ;
					qb = L - bh * nQ;
				}
			}
			const int q0 = qb * AQ;
			const f16* const Q = q + (long long)bh * T * HEAD_DIM;
			const f16* const K = k + (long long)bh * T * HEAD_DIM;
			const f16* const VT = vT + (long long)bh * HEAD_DIM * Tpad;
			const int keyBase = wave * 32 * KT;

			// Q tile -> LDS (one 16-byte chunk per thread)
			{
				const int row = tid >> 3;
				const int kc = ( tid & 7 ) * 8;
				int qr = q0 + row;
				qr = qr < T ? qr : T - 1;
				*(u32x4*)( ldsQ + row * QSTRIDE + kc ) = *(const u32x4*)( Q + (long long)qr * HEAD_DIM + kc );
			}
			__syncthreads();

			// ---- S^T = K . Q^T ----  (the K fragments of tile kt+1 are requested before the MFMAs of tile kt)
			f32x16 S[ KT ][ 2 ];
			f16x8 kf[ 4 ], kn[ 4 ];
			{
				int key = keyBase + c;
				key = key < T ? key : T - 1;
#pragma unroll
				for( int kk = 0; kk < 4; kk++ ) kf[ kk ] = *(const f16x8*)( K + (long long)key * HEAD_DIM + kk * 16 + hi * 8 );
			}
#pragma unroll
			for( int kt = 0; kt < KT; kt++ )
			{
				if( kt + 1 < KT )
				{
					int key = keyBase + ( kt + 1 ) * 32 + c;
					key = key < T ? key : T - 1;
#pragma unroll
					for( int kk = 0; kk < 4; kk++ ) kn[ kk ] = *(const f16x8*)( K + (long long)key * HEAD_DIM + kk * 16 + hi * 8 );
				}
#pragma unroll
				for( int qt = 0; qt < 2; qt++ )
				{
					f32x16 acc;
#pragma unroll
					for( int r = 0; r < 16; r++ ) acc[ r ] = 0.0f;
#pragma unroll
					for( int kk = 0; kk < 4; kk++ )
					{
						const f16x8 qf = *(const f16x8*)( ldsQ + ( qt * 32 + c ) * QSTRIDE + kk * 16 + hi * 8 );
						acc = __builtin_amdgcn_mfma_f32_32x32x16_f16( kf[ kk ], qf, acc, 0, 0, 0 );
					}
					S[ kt ][ qt ] = acc;
				}
				if( kt + 1 < KT )
				{
#pragma unroll
					for( int kk = 0; kk < 4; kk++ ) kf[ kk ] = kn[ kk ];
				}
			}

			// ---- scale, mask the padded keys, row maximum ----
			const float scale = 0.125f;	   // 1 / sqrt(64)
			float mx[ 2 ] = { -INFINITY, -INFINITY };
#pragma unroll
			for( int kt = 0; kt < KT; kt++ )
#pragma unroll
				for( int qt = 0; qt < 2; qt++ )
#pragma unroll
					for( int r = 0; r < 16; r++ )
					{
						const int key = keyBase + kt * 32 + ( r & 3 ) + 8 * ( r >> 2 ) + 4 * hi;
						float s = S[ kt ][ qt ][ r ] * scale;
						s = key < T ? s : -INFINITY;
						S[ kt ][ qt ][ r ] = s;
						mx[ qt ] = fmaxf( mx[ qt ], s );
					}
#pragma unroll
			for( int qt = 0; qt < 2; qt++ )
			{
				mx[ qt ] = fmaxf( mx[ qt ], __shfl_xor( mx[ qt ], 32, 64 ) );
				if( hi == 0 ) redMax[ wave * AQ + qt * 32 + c ] = mx[ qt ];
			}
			__syncthreads();
#pragma unroll
			for( int qt = 0; qt < 2; qt++ )
			{
				float m = redMax[ qt * 32 + c ];
#pragma unroll
				for( int w = 1; w < AWAVES; w++ ) m = fmaxf( m, redMax[ w * AQ + qt * 32 + c ] );
				mx[ qt ] = m;
			}

			// ---- e = exp16(S - max), row sums in double ----
			// per key tile the 16 values of a lane are FP16-exact numbers <= 1, so their FP32 sum is exact; tiles, lane
			// halves and waves are combined in double like the reference's row sum
			double sum[ 2 ] = { 0.0, 0.0 };
#pragma unroll
			for( int kt = 0; kt < KT; kt++ )
#pragma unroll
				for( int qt = 0; qt < 2; qt++ )
				{
					float part = 0.0f;
#pragma unroll
					for( int r = 0; r < 16; r++ )
					{
						const float s = S[ kt ][ qt ][ r ];
						const float e = ( s == -INFINITY ) ? 0.0f : exp16( s - mx[ qt ] );
						S[ kt ][ qt ][ r ] = e;
						part += e;
					}
					sum[ qt ] += (double)part;
				}
#pragma unroll
			for( int qt = 0; qt < 2; qt++ )
			{
				sum[ qt ] += __shfl_xor( sum[ qt ], 32, 64 );
				if( hi == 0 ) redSum[ wave * AQ + qt * 32 + c ] = sum[ qt ];
			}
			__syncthreads();
			float inv[ 2 ];
#pragma unroll
			for( int qt = 0; qt < 2; qt++ )
			{
				double t = redSum[ qt * 32 + c ];
#pragma unroll
				for( int w = 1; w < AWAVES; w++ ) t += redSum[ w * AQ + qt * 32 + c ];
				inv[ qt ] = (float)( 1.0 / t );
			}

			__builtin_amdgcn_sched_barrier( 0 );
			// ---- P = fp16(e * inv), packed in the B-operand layout of the P.V product ----
			f16x8 P[ KT ][ 2 ][ 2 ];	// [key tile][16-key step][query tile]
#pragma unroll
			for( int kt = 0; kt < KT; kt++ )
#pragma unroll
				for( int st = 0; st < 2; st++ )
#pragma unroll
					for( int qt = 0; qt < 2; qt++ )
#pragma unroll
						for( int j = 0; j < 8; j++ )
							P[ kt ][ st ][ qt ][ j ] = (f16)( S[ kt ][ qt ][ 8 * st + j ] * inv[ qt ] );

			// keep the phases apart for the register allocator: S (192 registers) must be fully packed into P (96) before the
			// 64 accumulators of O come alive
			__builtin_amdgcn_sched_barrier( 0 );

			// ---- O^T = V^T . P^T over this wave's keys ----
			f32x16 O[ 2 ][ 2 ];	   // [dd tile][query tile]
#pragma unroll
			for( int a = 0; a < 2; a++ )
#pragma unroll
				for( int b = 0; b < 2; b++ )
#pragma unroll
					for( int r = 0; r < 16; r++ ) O[ a ][ b ][ r ] = 0.0f;
#pragma unroll
			for( int kt = 0; kt < KT; kt++ )
#pragma unroll
				for( int st = 0; st < 2; st++ )
				{
					// fragment-major V (gemm.hip vFragIndex): one coalesced 16-byte load per lane and operand
					const int kb = ( keyBase + kt * 32 + 16 * st ) >> 4;
					f16x8 vf[ 2 ];
#pragma unroll
					for( int ddt = 0; ddt < 2; ddt++ )
						vf[ ddt ] = *(const f16x8*)( VT + ( ( (long long)kb * 2 + ddt ) * 64 + lane ) * 8 );
#pragma unroll
					for( int ddt = 0; ddt < 2; ddt++ )
#pragma unroll
						for( int qt = 0; qt < 2; qt++ )
							O[ ddt ][ qt ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( vf[ ddt ], P[ kt ][ st ][ qt ], O[ ddt ][ qt ], 0, 0, 0 );
				}

			// ---- deterministic tree reduction of the 8 partial tiles through LDS: (0+4) (1+5) (2+6) (3+7) -> (0+2) (1+3) -> 0+1
			// O^T layout: dd = ddt*32 + (r&3) + 8*(r>>2) + 4*hi (row), query = qt*32 + c (column); LDS image is [query][dd]
#pragma unroll
			for( int level = 4; level >= 1; level >>= 1 )
			{
				if( wave >= level && wave < 2 * level )
				{
					float* const dst = obuf + ( wave - level ) * AQ * OSTRIDE;
#pragma unroll
					for( int ddt = 0; ddt < 2; ddt++ )
#pragma unroll
						for( int qt = 0; qt < 2; qt++ )
#pragma unroll
							for( int r = 0; r < 16; r++ )
								dst[ ( qt * 32 + c ) * OSTRIDE + ddt * 32 + ( r & 3 ) + 8 * ( r >> 2 ) + 4 * hi ] = O[ ddt ][ qt ][ r ];
				}
				__syncthreads();
				if( wave < level )
				{
					const float* const src = obuf + wave * AQ * OSTRIDE;
#pragma unroll
					for( int ddt = 0; ddt < 2; ddt++ )
#pragma unroll
						for( int qt = 0; qt < 2; qt++ )
#pragma unroll
							for( int r = 0; r < 16; r++ )
								O[ ddt ][ qt ][ r ] += src[ ( qt * 32 + c ) * OSTRIDE + ddt * 32 + ( r & 3 ) + 8 * ( r >> 2 ) + 4 * hi ];
				}
				__syncthreads();
			}
			if( wave == 0 )
			{
#pragma unroll
				for( int ddt = 0; ddt < 2; ddt++ )
#pragma unroll
					for( int qt = 0; qt < 2; qt++ )
#pragma unroll
						for( int r = 0; r < 16; r++ )
							obuf[ ( qt * 32 + c ) * OSTRIDE + ddt * 32 + ( r & 3 ) + 8 * ( r >> 2 ) + 4 * hi ] = O[ ddt ][ qt ][ r ];
			}
			__syncthreads();

			// ---- store: out[b][t][h*64 + dd] FP16 (KQV_merged, whisper.cpp:1317-1321; the next product rounds it anyway)
			{
				const int row = tid >> 3;
				const int d0 = ( tid & 7 ) * 8;
				const int t = q0 + row;
				if( t < T )
				{
					const int b = bh / heads, h = bh - b * heads;
					f16x8 pk;
#pragma unroll
					for( int j = 0; j < 8; j++ ) pk[ j ] = (f16)obuf[ row * OSTRIDE + d0 + j ];
					*(f16x8*)( out + ( (long long)b * T + t ) * ( heads * HEAD_DIM ) + h * HEAD_DIM + d0 ) = pk;
				}
			}
		}

		// ---------------------------------------------------------------------------------------------------------------
		// attentionEncF: the same arithmetic (exact ggml_flash_attn_f16 semantics: P rounded to FP16 AFTER normalising by the
		// full row sum) with the scores RECOMPUTED instead of kept: three sweeps over the keys -- row maximum, row sum of
		// exp16(s - max), then P.V -- each forming S^T = K.Q^T again on the matrix cores. That doubles the MFMA work of the
		// product but removes what made the first kernel slow: 192 score registers per lane (one workgroup per CU, every
		// wave in the same phase), K/V read from L2 by each 64-query workgroup, and a three-level cross-wave reduction of O.
		//   * a wave owns 32 query rows and ALL keys: no cross-wave exchange at all (max / sum / O are per lane + one xor-32 shuffle);
		//   * 8 waves = 256 query rows share every K tile (128 keys x 64 dims, XOR-swizzled like the GEMM's A tile) and V tile
		//     (fragment-major, lane-linear) through LDS, filled by global_load_lds_dwordx4, double buffered, one barrier per tile;
		//   * ~110 registers: two workgroups per CU, so one's exp / convert VALU work runs under the other's MFMAs.
		// S is bit-identical in the three sweeps (same instruction sequence), so max, sum and P are consistent.
		constexpr int FQ = 256;					   // query rows per workgroup (8 waves x 32)
		constexpr int FK = 128;					   // keys per LDS tile
		constexpr int F_TILE = FK * HEAD_DIM;	   // halfs per tile (16 KiB)
		constexpr int F_LDS_BYTES = 4 * F_TILE * 2;   // K and V, two buffers each

		// TWO: the second and third sweep are one -- O accumulates V^T . e^T with e = exp16( s - max ) (an FP16 number already, so
		// the operand is exact) while the row sum is formed, and O is scaled by 1 / sum at the end in FP32. That drops the
		// reference's rounding of e / sum to FP16 (ggml.c:6035-6046), i.e. it is closer to the exact softmax than the
		// reference is, not bit-compatible with it; the kernel is bound by the exponentials (VALU), and this halves them.
		// (A one-multiply exponential -- exp2( fp16( x ) * log2 e ) without the hi/lo split of exp16 -- measured SLOWER in this
		// kernel, 2782 vs 2457 us per launch at 112 windows, and is not table-exact: retired.)
		template<bool TWO>
		__global__ void __launch_bounds__( 512, 4 ) attentionEncF( const f16* __restrict__ q, const f16* __restrict__ k,
			const f16* __restrict__ vT, f16* __restrict__ out, int heads, int T, int Tpad, int nQ, int xcdRemap )
		{
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemF[];
			f16* const ldsK = (f16*)smemF;				  // [2][128][64], chunk-swizzled rows
			f16* const ldsV = ldsK + 2 * F_TILE;		  // [2][8 key blocks][2 dd halves][64 lanes][8]
			typedef __attribute__( ( address_space( 3 ) ) ) void* LdsPtr;
			typedef const __attribute__( ( address_space( 1 ) ) ) void* GlobalPtr;

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = tid >> 6;
			const int hi = lane >> 5;
			const int c = lane & 31;
			int bh, qb;
			{
				const int L = blockIdx.x;
				if( xcdRemap )
				{
					const int kIdx = L >> 3;
					bh = ( L & 7 ) + 8 * ( kIdx / nQ );
					qb = kIdx % nQ;
				}
				else
				{
					bh = L / nQ;
					qb = L - bh * nQ;
				}
			}
			const f16* const Q = q + (long long)bh * T * HEAD_DIM;
			const f16* const K = k + (long long)bh * T * HEAD_DIM;
			const f16* const VT = vT + (long long)bh * HEAD_DIM * Tpad;
			const int qRow = qb * FQ + wave * 32 + c;
			const int nTiles = ( T + FK - 1 ) / FK;

			// Q fragments of this lane's query row (B operand), kept for the whole kernel, pre-multiplied by 1 / sqrt(64) = 2^-3: exact
			// (bar FP16 underflow of |q| < 5e-4, which moves a score by < 1e-8), and S then needs no multiply per element
			f16x8 qf[ 4 ];
			{
				const int qr = qRow < T ? qRow : T - 1;
	#pragma unroll
				for( int kk = 0; kk < 4; kk++ )
				{
					qf[ kk ] = *(const f16x8*)( Q + (long long)qr * HEAD_DIM + kk * 16 + hi * 8 );
	#pragma unroll
					for( int j = 0; j < 8; j++ ) qf[ kk ][ j ] = qf[ kk ][ j ] * (f16)0.125f;
				}
			}

			// K tile t -> LDS buffer: two 1 KiB wave instructions per wave (8 rows each); the XOR of the 16-byte chunk index is
			// applied to the SOURCE address, the LDS image of an instruction is lane-linear
			auto issueK = [ & ]( int t, int buf )
			{
	#pragma unroll
				for( int i = 0; i < 2; i++ )
				{
					const int row = ( wave * 2 + i ) * 8 + ( lane >> 3 );
					const int cl = ( lane & 7 ) ^ ( ( row >> 1 ) & 7 );
					int key = t * FK + row;
					key = key < T ? key : T - 1;
					__builtin_amdgcn_global_load_lds( (GlobalPtr)( K + (long long)key * HEAD_DIM + cl * 8 ),
						(LdsPtr)( ldsK + buf * F_TILE + ( wave * 2 + i ) * 512 ), 16, 0, 0 );
				}
			};
			auto issueV = [ & ]( int t, int buf )
			{
	#pragma unroll
				for( int i = 0; i < 2; i++ )
					__builtin_amdgcn_global_load_lds( (GlobalPtr)( VT + (long long)t * F_TILE + ( wave * 2 + i ) * 512 + lane * 8 ),
						(LdsPtr)( ldsV + buf * F_TILE + ( wave * 2 + i ) * 512 ), 16, 0, 0 );
			};
			// S^T of one 32-key sub-tile (already scaled): keys (r & 3) + 8 (r >> 2) + 4 hi down the registers, this lane's query across.
			// The accumulator starts from the constant 0 (an inline operand of the first MFMA, no register initialisation).
			// Keys >= T (only the last tile has any) are overwritten with -3e38 afterwards: the row maximum ignores them and the
			// clamp of the exponential's argument turns them into exp16( -64 ) = 0.
			auto scores = [ & ]( const f16* kt, int st, int t ) -> f32x16
			{
				f32x16 acc;
	#pragma unroll
				for( int r = 0; r < 16; r++ ) acc[ r ] = 0.0f;
				const int row = st * 32 + c;
				const int sw = ( row >> 1 ) & 7;
	#pragma unroll
				for( int kk = 0; kk < 4; kk++ )
				{
					const f16x8 kf = *(const f16x8*)( kt + row * HEAD_DIM + ( ( ( kk * 2 + hi ) ^ sw ) << 3 ) );
					acc = __builtin_amdgcn_mfma_f32_32x32x16_f16( kf, qf[ kk ], acc, 0, 0, 0 );
				}
				if( t == nTiles - 1 )
				{
					// opaque to the optimiser: otherwise the 4 x 16 mask tests of the last tile are hoisted out of the sweeps and
					// live in registers for the whole kernel (608 spilled VGPRs)
					int limit = T - ( t * FK + st * 32 + 4 * hi );
					asm volatile( "" : "+v"( limit ) );
	#pragma unroll
					for( int r = 0; r < 16; r++ ) acc[ r ] = ( r & 3 ) + 8 * ( r >> 2 ) < limit ? acc[ r ] : -3.0e38f;
				}
				return acc;
			};
			auto sweepTile = [ & ]( int t, auto&& body )
			{
	#pragma unroll
				for( int st = 0; st < 4; st++ ) body( st );
			};
			// exp16 of a score: the argument is clamped to [-64, 0] (exp16( -64 ) == 0 == exp16 of anything below -17.4, and a real
			// key never exceeds the row maximum), which also absorbs the padded keys' -3e38
			auto expScore = [ & ]( float sc, float mxv ) -> float { return exp16( __builtin_amdgcn_fmed3f( sc - mxv, -64.0f, 0.0f ) ); };
			// sum of 16 FP16-exact numbers <= 1 (the e of one sub-tile, packed as the P operand): v_dot2_f32_f16 against ones adds two
			// per instruction in FP32 without converting them back first
			auto sumP = [ & ]( const f16x8 ( &P )[ 2 ] ) -> float
			{
				typedef _Float16 h2 __attribute__( ( ext_vector_type( 2 ) ) );
				const h2 ones = { (f16)1.0f, (f16)1.0f };
				float part = 0.0f;
	#pragma unroll
				for( int h = 0; h < 2; h++ )
	#pragma unroll
					for( int j = 0; j < 8; j += 2 )
						part = __builtin_amdgcn_fdot2( h2{ P[ h ][ j ], P[ h ][ j + 1 ] }, ones, part, false );
				return part;
			};

			// ---- sweep 1: row maximum ----
			float mx = -INFINITY;
			issueK( 0, 0 );
			for( int t = 0; t < nTiles; t++ )
			{
				const int buf = t & 1;
				asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
				__syncthreads();
				if( t + 1 < nTiles ) issueK( t + 1, buf ^ 1 );
				const f16* const kt = ldsK + buf * F_TILE;
				sweepTile( t, [ & ]( int st )
				{
					const f32x16 S = scores( kt, st, t );
	#pragma unroll
					for( int r = 0; r < 16; r++ ) mx = fmaxf( mx, S[ r ] );
				} );
			}
			mx = fmaxf( mx, __shfl_xor( mx, 32, 64 ) );

			// ---- sweep 2: row sum of exp16( s - max ), per-lane FP32 partials of a sub-tile combined in double ----
			double sum = 0.0;
			if constexpr( !TWO )
			{
			__syncthreads();
			issueK( 0, 0 );
			for( int t = 0; t < nTiles; t++ )
			{
				const int buf = t & 1;
				asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
				__syncthreads();
				if( t + 1 < nTiles ) issueK( t + 1, buf ^ 1 );
				const f16* const kt = ldsK + buf * F_TILE;
				sweepTile( t, [ & ]( int st )
				{
					const f32x16 S = scores( kt, st, t );
					f16x8 E[ 2 ];
	#pragma unroll
					for( int r = 0; r < 16; r++ ) E[ r >> 3 ][ r & 7 ] = (f16)expScore( S[ r ], mx );
					sum += (double)sumP( E );
				} );
			}
			sum += __shfl_xor( sum, 32, 64 );
			}
			const float inv = TWO ? 1.0f : (float)( 1.0 / sum );

			// ---- sweep 3: P = fp16( e * inv ) packed straight into the B operand of O^T += V^T . P^T ----
			f32x16 O[ 2 ];
	#pragma unroll
			for( int a = 0; a < 2; a++ )
	#pragma unroll
				for( int r = 0; r < 16; r++ ) O[ a ][ r ] = 0.0f;
			__syncthreads();
			issueK( 0, 0 );
			issueV( 0, 0 );
			for( int t = 0; t < nTiles; t++ )
			{
				const int buf = t & 1;
				asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
				__syncthreads();
				if( t + 1 < nTiles )
				{
					issueK( t + 1, buf ^ 1 );
					issueV( t + 1, buf ^ 1 );
				}
				const f16* const kt = ldsK + buf * F_TILE;
				const f16* const vt = ldsV + buf * F_TILE;
				sweepTile( t, [ & ]( int st )
				{
					const f32x16 S = scores( kt, st, t );
					f16x8 P[ 2 ];
					if constexpr( TWO )
					{
	#pragma unroll
						for( int r = 0; r < 16; r++ ) P[ r >> 3 ][ r & 7 ] = (f16)expScore( S[ r ], mx );
						sum += (double)sumP( P );
					}
					else
					{
	#pragma unroll
						for( int r = 0; r < 16; r++ ) P[ r >> 3 ][ r & 7 ] = (f16)( expScore( S[ r ], mx ) * inv );
					}
	#pragma unroll
					for( int half = 0; half < 2; half++ )
					{
						const int kb = st * 2 + half;	  // 16-key block inside the tile
	#pragma unroll
						for( int ddt = 0; ddt < 2; ddt++ )
						{
							const f16x8 vf = *(const f16x8*)( vt + ( ( kb * 2 + ddt ) * 64 + lane ) * 8 );
							O[ ddt ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( vf, P[ half ], O[ ddt ], 0, 0, 0 );
						}
					}
				} );
			}

			if constexpr( TWO )
			{
				sum += __shfl_xor( sum, 32, 64 );
				const float invSum = (float)( 1.0 / sum );
	#pragma unroll
				for( int a = 0; a < 2; a++ )
	#pragma unroll
					for( int r = 0; r < 16; r++ ) O[ a ][ r ] *= invSum;
			}
			// ---- store: out[b][t][h*64 + dd] FP16; O^T rows are dd = ddt*32 + (r&3) + 8 (r>>2) + 4 hi, the column is this lane's query
			if( qRow < T )
			{
				const int b = bh / heads, h = bh - b * heads;
				f16* const o = out + ( (long long)b * T + qRow ) * ( heads * HEAD_DIM ) + h * HEAD_DIM;
	#pragma unroll
				for( int ddt = 0; ddt < 2; ddt++ )
	#pragma unroll
					for( int g4 = 0; g4 < 4; g4++ )
					{
						f16x4 pk;
	#pragma unroll
						for( int e = 0; e < 4; e++ ) pk[ e ] = (f16)O[ ddt ][ 4 * g4 + e ];
						*(f16x4*)( o + ddt * 32 + 8 * g4 + 4 * hi ) = pk;
					}
			}
		}


		// ---------------------------------------------------------------------------------------------------------------
		// attentionEncT: attentionEncF<TWO = true> with the exponential as what it is in the reference -- a LOOKUP in table_exp_f16
		// (Whisper/source/ggml.c:1375-1385; softmax of flash_attn_f16 :6001-6016). attentionEncF is bound by the VALU: its table-exact
		// exp16 costs 14 issue slots per score (two converts, the hi / lo split of x log2 e, v_exp_f32 at quarter rate, the correction,
		// the FP16 rounding) against 0.19 matrix-core cycles. Here a score costs
		//     v_sub (s - max), v_cvt_pk_f16_f32 (two scores), and / v_pk_min_u16 (|bits| clamped to the first zero entry), two shifts
		//     = 4 VALU slots, and ONE ds_read_u16 from the model's table in LDS (kernels.h EXP_TABLE_ENTRIES: the non-positive
		//     arguments' 20480 entries, 40 KB) -- the LDS pipe is idle otherwise (4 fragment reads per 1024 scores).
		// The result is the reference's e bit for bit (exp16 differs from the table in ~1e-4 of the inputs).
		// 16 waves = 512 query rows share the K / V tiles AND the table: 104 KB of LDS, one 1024-thread workgroup per CU (the same
		// 16 waves per CU attentionEncF runs as two workgroups).
		constexpr int TQ = 512;
		constexpr int T_TABLE_HALFS = EXP_TABLE_ENTRIES;
		constexpr int T_LDS_BYTES = T_TABLE_HALFS * 2 + 4 * F_TILE * 2;

		// MODE 0: the table lookup above (bit-exact e; the LDS pipe is its bound: 16 two-byte gathers per 32 x 32 sub-tile and wave, ~4100 cycles per sub-tile
		//         and CU against ~1000 on the matrix cores -- round 5's 0.22 of the MFMA peak).
		// MODE 1 (round 6, the timed path): e = fp16( 2^( fp32( fp16( s - max ) ) * log2 e ) ) on the VALU -- v_exp_f32 issues at ~5/3 of a plain VALU slot on
		//         gfx950 (MI355X_MICROARCH.md), so a score costs ~4.2 slots and no LDS access: the sub-tile's exponentials (~1070 cycles per SIMD) balance its
		//         8 MFMAs (1024). Same argument rounding as the reference (fp16( s - max ), ggml.c:6008); the FP32 product x * log2 e and v_exp_f32's last
		//         bit move e across an FP16 rounding boundary in ~0.3 % of the entries (one FP16 ulp each) -- measured against the table kernel in
		//         tests/test_gpu_ops.py::test_encoder_attention_valu_exp.
		template<int MODE>
		__global__ void __launch_bounds__( 1024, 4 ) attentionEncT( const f16* __restrict__ q, const f16* __restrict__ k,
			const f16* __restrict__ vT, f16* __restrict__ out, const f16* __restrict__ expTab, int heads, int T, int Tpad, int nQ, int xcdRemap )
		{
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemT[];
			f16* const ldsTab = (f16*)smemT;				  // [EXP_TABLE_ENTRIES] at LDS offset 0 (MODE 0 only)
			f16* const ldsK = ldsTab + ( MODE == 0 ? T_TABLE_HALFS : 0 );		  // [2][128][64], chunk-swizzled rows
			f16* const ldsV = ldsK + 2 * F_TILE;			  // [2][8 key blocks][2 dd halves][64 lanes][8]
			typedef __attribute__( ( address_space( 3 ) ) ) void* LdsPtr;
			typedef const __attribute__( ( address_space( 1 ) ) ) void* GlobalPtr;

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = tid >> 6;
			const int hi = lane >> 5;
			const int c = lane & 31;
			int bh, qb;
			{
				const int L = blockIdx.x;
				if( xcdRemap )
				{
					const int kIdx = L >> 3;
					bh = ( L & 7 ) + 8 * ( kIdx / nQ );
					qb = kIdx % nQ;
				}
				else
				{
					bh = L / nQ;
					qb = L - bh * nQ;
				}
			}
			const f16* const Q = q + (long long)bh * T * HEAD_DIM;
			const f16* const K = k + (long long)bh * T * HEAD_DIM;
			const f16* const VT = vT + (long long)bh * HEAD_DIM * Tpad;
			const int qRow = qb * TQ + wave * 32 + c;
			const int nTiles = ( T + FK - 1 ) / FK;

			// the table: 40 pieces of 1 KiB, lane-linear; needed from the second sweep on, so it lands under the first
			if constexpr( MODE == 0 )
			for( int piece = wave; piece < T_TABLE_HALFS / 512; piece += 16 )
				__builtin_amdgcn_global_load_lds( (GlobalPtr)( expTab + piece * 512 + lane * 8 ), (LdsPtr)( ldsTab + piece * 512 ), 16, 0, 0 );

			f16x8 qf[ 4 ];
			{
				const int qr = qRow < T ? qRow : T - 1;
	#pragma unroll
				for( int kk = 0; kk < 4; kk++ )
				{
					qf[ kk ] = *(const f16x8*)( Q + (long long)qr * HEAD_DIM + kk * 16 + hi * 8 );
	#pragma unroll
					for( int j = 0; j < 8; j++ ) qf[ kk ][ j ] = qf[ kk ][ j ] * (f16)0.125f;
				}
			}
			// K tile t -> LDS buffer: 16 pieces of 8 rows, one 1 KiB wave instruction per wave (swizzle on the SOURCE address)
			auto issueK = [ & ]( int t, int buf )
			{
				const int row = wave * 8 + ( lane >> 3 );
				const int cl = ( lane & 7 ) ^ ( ( row >> 1 ) & 7 );
				int key = t * FK + row;
				key = key < T ? key : T - 1;
				__builtin_amdgcn_global_load_lds( (GlobalPtr)( K + (long long)key * HEAD_DIM + cl * 8 ), (LdsPtr)( ldsK + buf * F_TILE + wave * 512 ), 16, 0, 0 );
			};
			auto issueV = [ & ]( int t, int buf )
			{
				__builtin_amdgcn_global_load_lds( (GlobalPtr)( VT + (long long)t * F_TILE + wave * 512 + lane * 8 ), (LdsPtr)( ldsV + buf * F_TILE + wave * 512 ), 16, 0, 0 );
			};
			auto scores = [ & ]( const f16* kt, int st, int t ) -> f32x16
			{
				f32x16 acc;
	#pragma unroll
				for( int r = 0; r < 16; r++ ) acc[ r ] = 0.0f;
				const int row = st * 32 + c;
				const int sw = ( row >> 1 ) & 7;
	#pragma unroll
				for( int kk = 0; kk < 4; kk++ )
				{
					const f16x8 kf = *(const f16x8*)( kt + row * HEAD_DIM + ( ( ( kk * 2 + hi ) ^ sw ) << 3 ) );
					acc = __builtin_amdgcn_mfma_f32_32x32x16_f16( kf, qf[ kk ], acc, 0, 0, 0 );
				}
				if( t == nTiles - 1 )
				{
					// (peeling the last tile into its own instantiation, so that the other tiles carry no compares / selects, was measured in round 6: the second copy of
					// the loop body costs 17 spilled VGPRs under the 128-register cap and the kernel 8 %: profiles/r06_evidence/enc_attn_ablation.txt)
					int limit = T - ( t * FK + st * 32 + 4 * hi );
					asm volatile( "" : "+v"( limit ) );
	#pragma unroll
					for( int r = 0; r < 16; r++ ) acc[ r ] = ( r & 3 ) + 8 * ( r >> 2 ) < limit ? acc[ r ] : -3.0e38f;	  // fp16( -3e38 - max ) = -inf -> the zero entry
				}
				return acc;
			};
			auto sumP = [ & ]( const f16x8 ( &P )[ 2 ] ) -> float
			{
				typedef _Float16 h2 __attribute__( ( ext_vector_type( 2 ) ) );
				const h2 ones = { (f16)1.0f, (f16)1.0f };
				float part = 0.0f;
	#pragma unroll
				for( int h = 0; h < 2; h++ )
	#pragma unroll
					for( int j = 0; j < 8; j += 2 )
						part = __builtin_amdgcn_fdot2( h2{ P[ h ][ j ], P[ h ][ j + 1 ] }, ones, part, false );
				return part;
			};

			// ---- sweep 1: row maximum ----
			float mx = -INFINITY;
			issueK( 0, 0 );
			for( int t = 0; t < nTiles; t++ )
			{
				const int buf = t & 1;
				asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
				__syncthreads();
				if( t + 1 < nTiles ) issueK( t + 1, buf ^ 1 );
				const f16* const kt = ldsK + buf * F_TILE;
	#pragma unroll
				for( int st = 0; st < 4; st++ )
				{
					const f32x16 S = scores( kt, st, t );
	#pragma unroll
					for( int r = 0; r < 16; r++ ) mx = fmaxf( mx, S[ r ] );
				}
			}
			mx = fmaxf( mx, __shfl_xor( mx, 32, 64 ) );

			// ---- sweep 2: e = table[ |fp16( s - max )| ] straight into the B operand of O^T += V^T . e^T, row sum on the way ----
			double sum = 0.0;
			f32x16 O[ 2 ];
	#pragma unroll
			for( int a = 0; a < 2; a++ )
	#pragma unroll
				for( int r = 0; r < 16; r++ ) O[ a ][ r ] = 0.0f;
			__syncthreads();
			issueK( 0, 0 );
			issueV( 0, 0 );
			for( int t = 0; t < nTiles; t++ )
			{
				const int buf = t & 1;
				asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
				__syncthreads();
				if( t + 1 < nTiles )
				{
					issueK( t + 1, buf ^ 1 );
					issueV( t + 1, buf ^ 1 );
				}
				const f16* const kt = ldsK + buf * F_TILE;
				const f16* const vt = ldsV + buf * F_TILE;
	#pragma unroll
				for( int st = 0; st < 4; st++ )
				{
					const f32x16 S = scores( kt, st, t );
					f16x8 P[ 2 ];
	#pragma unroll
					for( int r = 0; r < 16; r += 2 )
					{
						typedef _Float16 h2 __attribute__( ( ext_vector_type( 2 ) ) );
						typedef unsigned short u16x2 __attribute__( ( ext_vector_type( 2 ) ) );
						const h2 hd = { (f16)( S[ r ] - mx ), (f16)( S[ r + 1 ] - mx ) };	  // the reference's fp16( s - max ) (ggml.c:6008)
						if constexpr( MODE == 0 )
						{
							u16x2 mag = __builtin_bit_cast( u16x2, __builtin_bit_cast( unsigned, hd ) & 0x7FFF7FFFu );
							const u16x2 lim = { (unsigned short)EXP_TABLE_LIMIT, (unsigned short)EXP_TABLE_LIMIT };
							mag = __builtin_elementwise_min( mag, lim );
							P[ r >> 3 ][ r & 7 ] = ldsTab[ mag[ 0 ] ];
							P[ r >> 3 ][ ( r & 7 ) + 1 ] = ldsTab[ mag[ 1 ] ];
						}
						else
						{
							typedef float f2 __attribute__( ( ext_vector_type( 2 ) ) );
							constexpr float L2E = 1.44269504088896340736f;
							const h2 hp = __builtin_convertvector( f2{ S[ r ] - mx, S[ r + 1 ] - mx }, h2 );	   // one v_cvt_pk_f16_f32
							const f2 e = { __builtin_amdgcn_exp2f( __builtin_fmaf( (float)hp[ 0 ], L2E, 0.0f ) ), __builtin_amdgcn_exp2f( __builtin_fmaf( (float)hp[ 1 ], L2E, 0.0f ) ) };	   // v_fma_mix_f32 + v_exp_f32
							const h2 eh = __builtin_convertvector( e, h2 );
							P[ r >> 3 ][ r & 7 ] = eh[ 0 ];
							P[ r >> 3 ][ ( r & 7 ) + 1 ] = eh[ 1 ];
						}
					}
					sum += (double)sumP( P );
	#pragma unroll
					for( int half = 0; half < 2; half++ )
					{
						const int kb = st * 2 + half;
	#pragma unroll
						for( int ddt = 0; ddt < 2; ddt++ )
						{
							const f16x8 vf = *(const f16x8*)( vt + ( ( kb * 2 + ddt ) * 64 + lane ) * 8 );
							O[ ddt ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( vf, P[ half ], O[ ddt ], 0, 0, 0 );
						}
					}
				}
			}
			sum += __shfl_xor( sum, 32, 64 );
			const float invSum = (float)( 1.0 / sum );
	#pragma unroll
			for( int a = 0; a < 2; a++ )
	#pragma unroll
				for( int r = 0; r < 16; r++ ) O[ a ][ r ] *= invSum;
			if( qRow < T )
			{
				const int b = bh / heads, h = bh - b * heads;
				f16* const o = out + ( (long long)b * T + qRow ) * ( heads * HEAD_DIM ) + h * HEAD_DIM;
	#pragma unroll
				for( int ddt = 0; ddt < 2; ddt++ )
	#pragma unroll
					for( int g4 = 0; g4 < 4; g4++ )
					{
						f16x4 pk;
	#pragma unroll
						for( int e = 0; e < 4; e++ ) pk[ e ] = (f16)O[ ddt ][ 4 * g4 + e ];
						*(f16x4*)( o + ddt * 32 + 8 * g4 + 4 * hi ) = pk;
					}
			}
		}

		// ---------------------------------------------------------------------------------------------------------------
		// attentionEncW (round 6, the timed path): attentionEncT's layout -- a wave owns 32 x NG query rows and all keys, 512 rows per workgroup share the K / V
		// tiles through LDS -- rebuilt around what the counters say binds it (profiles/r06_evidence/enc_attn_ablation.txt, enc_attn_pmc_mode2.json): the VALU.
		// attentionEncT spends ~780 VALU instructions per 128-key tile and wave against 48 MFMAs (1536 matrix-pipe cycles): 3120 issue cycles x 4 waves per SIMD =
		// twice the matrix time. Of those, 258 are the compare / select pairs that mask keys >= T -- needed in the LAST tile only, executed in every tile once the
		// compiler turns `if( t == nTiles - 1 )` into selects --, ~64 zero the score accumulators, and the exponential's conversions are one instruction per element.
		// Here:
		//   * the last tile is its own instantiation of the tile body (std::true_type / false_type): no masking code in the other eleven;
		//   * the first MFMA of a score tile takes the inline constant 0 as its C operand;
		//   * per TWO scores: 2 x v_sub, ONE v_cvt_pk_f16_f32 (the reference's fp16( s - max ), ggml.c:6008), 2 x v_fma_mix_f32 (FP16 source x log2 e),
		//     2 x v_exp_f32, ONE v_cvt_pk_f16_f32, one v_dot2c for the row sum: 4.5 VALU slots per score (attentionEncT<1>: ~7, the table kernel 4 + an LDS gather);
		//   * e = fp16( 2^( fp32( fp16( s - max ) ) * log2 e ) ): the FP32 product and v_exp_f32's last bit move e across an FP16 rounding boundary in ~0.3 % of
		//     the entries against the reference's table (one FP16 ulp each);
		//   * a ring of W_NBUF K / V tiles with counted waits (the loads of tile t + W_NBUF - 1 go out while tile t is computed).
		// NG = 1: 16 waves x 32 rows (4 waves per SIMD: the other waves' MFMAs cover a wave's exponentials and LDS latencies). NG = 2: 8 waves x 64 rows -- every
		// K / V fragment read from LDS feeds two MFMAs (half the LDS traffic per FLOP), but at 2 waves per SIMD the exposed ds_read / MFMA latencies cost what that saves
		// (measured equal, 1.87 vs 1.85 ms before the VALU diet); kept as an instantiation for the comparison.
		// ONLINE: ONE sweep -- the running maximum m of a query row is raised only when a sub-tile exceeds it by more than W_LAZY (then O and the row sum are
		// scaled by exp( m_old - m_new )), so the exponentials' arguments stay <= W_LAZY: e <= e^W_LAZY fits FP16 with the same relative precision, the FP32
		// sums absorb the range, and the rescale is rare (the first sub-tiles of a row). That removes the max sweep: a third of the MFMAs and K-tile reads. It is NOT
		// the reference's fp16( s - max ) argument any more (the argument is rounded against the running maximum): an FP16-rounding-sized change per e, held to
		// account by tests/test_gpu_ops.py::test_flash_attention and the exact-mode bounds of tests/test_gpu_exact.py.
		constexpr int W_NBUF = 4;
		constexpr int W_LDS_BYTES = 2 * W_NBUF * F_TILE * 2;
		constexpr float W_LAZY = 4.0f;
		template<int NG, bool ONLINE, bool NOARG = false>
		__global__ void __launch_bounds__( 1024 / NG, NG == 1 ? 4 : 2 ) attentionEncW( const f16* __restrict__ q, const f16* __restrict__ k,
			const f16* __restrict__ vT, f16* __restrict__ out, int heads, int T, int Tpad, int nQ, int xcdRemap )
		{
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemW[];
			f16* const ldsK = (f16*)smemW;				  // [W_NBUF][128][64], chunk-swizzled rows
			f16* const ldsV = ldsK + W_NBUF * F_TILE;	  // [W_NBUF][8 key blocks][2 dd halves][64 lanes][8]
			typedef __attribute__( ( address_space( 3 ) ) ) void* LdsPtr;
			typedef const __attribute__( ( address_space( 1 ) ) ) void* GlobalPtr;
			typedef _Float16 h2 __attribute__( ( ext_vector_type( 2 ) ) );
			typedef float f2 __attribute__( ( ext_vector_type( 2 ) ) );
			constexpr float L2E = 1.44269504088896340736f;

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = tid >> 6;
			const int hi = lane >> 5;
			const int c = lane & 31;
			int bh, qb;
			{
				const int L = blockIdx.x;
				if( xcdRemap )
				{
					const int kIdx = L >> 3;
					bh = ( L & 7 ) + 8 * ( kIdx / nQ );
					qb = kIdx % nQ;
				}
				else
				{
					bh = L / nQ;
					qb = L - bh * nQ;
				}
			}
			const f16* const Q = q + (long long)bh * T * HEAD_DIM;
			const f16* const K = k + (long long)bh * T * HEAD_DIM;
			const f16* const VT = vT + (long long)bh * HEAD_DIM * Tpad;
			const int nTiles = ( T + FK - 1 ) / FK;
			int qRow[ NG ];
			f16x8 qf[ NG ][ 4 ];
	#pragma unroll
			for( int g = 0; g < NG; g++ )
			{
				qRow[ g ] = qb * TQ + ( wave * NG + g ) * 32 + c;
				const int qr = qRow[ g ] < T ? qRow[ g ] : T - 1;
	#pragma unroll
				for( int kk = 0; kk < 4; kk++ )
				{
					qf[ g ][ kk ] = *(const f16x8*)( Q + (long long)qr * HEAD_DIM + kk * 16 + hi * 8 );
	#pragma unroll
					for( int j = 0; j < 8; j++ ) qf[ g ][ kk ][ j ] = qf[ g ][ kk ][ j ] * (f16)0.125f;
				}
			}
			// a tile = 16 pieces of 1 KiB; a wave issues NG of them per operand
			auto issueK = [ & ]( int t, int buf )
			{
	#pragma unroll
				for( int i = 0; i < NG; i++ )
				{
					const int row = ( wave * NG + i ) * 8 + ( lane >> 3 );
					const int cl = ( lane & 7 ) ^ ( ( row >> 1 ) & 7 );
					int key = t * FK + row;
					key = key < T ? key : T - 1;
					__builtin_amdgcn_global_load_lds( (GlobalPtr)( K + (long long)key * HEAD_DIM + cl * 8 ),
						(LdsPtr)( ldsK + buf * F_TILE + ( wave * NG + i ) * 512 ), 16, 0, 0 );
				}
			};
			auto issueV = [ & ]( int t, int buf )
			{
	#pragma unroll
				for( int i = 0; i < NG; i++ )
					__builtin_amdgcn_global_load_lds( (GlobalPtr)( VT + (long long)t * F_TILE + ( wave * NG + i ) * 512 + lane * 8 ),
						(LdsPtr)( ldsV + buf * F_TILE + ( wave * NG + i ) * 512 ), 16, 0, 0 );
			};
			// tile t has landed when at most the loads of the younger tiles (`per` instructions each) are outstanding; vmcnt takes an immediate
			auto waitTile = [ & ]( int t, auto per )
			{
				constexpr int P = decltype( per )::value;	   // load instructions per tile and wave
				const int younger = min( W_NBUF - 2, nTiles - 1 - t );
				if( younger >= 2 )
				{
					if constexpr( P == 2 ) asm volatile( "s_waitcnt vmcnt(4)" ::: "memory" );
					else asm volatile( "s_waitcnt vmcnt(8)" ::: "memory" );
				}
				else if( younger == 1 )
				{
					if constexpr( P == 2 ) asm volatile( "s_waitcnt vmcnt(2)" ::: "memory" );
					else asm volatile( "s_waitcnt vmcnt(4)" ::: "memory" );
				}
				else asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
			};
			static_assert( W_NBUF == 4 && NG == 2, "waitTile counts two younger tiles of 2 or 4 load instructions" );
			const f32x16 zero16 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
			// S^T of one 32-key sub-tile for the wave's query groups from ONE set of K fragments
			auto scores = [ & ]( const f16* kt, int st, int t, auto last, f32x16 ( &S )[ NG ] )
			{
				const int row = st * 32 + c;
				const int sw = ( row >> 1 ) & 7;
	#pragma unroll
				for( int kk = 0; kk < 4; kk++ )
				{
					const f16x8 kf = *(const f16x8*)( kt + row * HEAD_DIM + ( ( ( kk * 2 + hi ) ^ sw ) << 3 ) );
	#pragma unroll
					for( int g = 0; g < NG; g++ ) S[ g ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( kf, qf[ g ][ kk ], kk == 0 ? zero16 : S[ g ], 0, 0, 0 );
				}
				if constexpr( decltype( last )::value )
				{
					const int limit = T - ( t * FK + st * 32 + 4 * hi );
	#pragma unroll
					for( int r = 0; r < 16; r++ )
					{
						const bool in = ( r & 3 ) + 8 * ( r >> 2 ) < limit;
	#pragma unroll
						for( int g = 0; g < NG; g++ ) S[ g ][ r ] = in ? S[ g ][ r ] : -3.0e38f;	  // fp16( -3e38 - max ) = -inf -> e = 0
					}
				}
			};
			auto sumP = [ & ]( const f16x8 ( &P )[ 2 ] ) -> float
			{
				const h2 ones = { (f16)1.0f, (f16)1.0f };
				float part = 0.0f;
	#pragma unroll
				for( int h = 0; h < 2; h++ )
	#pragma unroll
					for( int j = 0; j < 8; j += 2 )
						part = __builtin_amdgcn_fdot2( h2{ P[ h ][ j ], P[ h ][ j + 1 ] }, ones, part, false );
				return part;
			};
			auto expTile = [ & ]( const f32x16& S, float mref, f16x8 ( &P )[ 2 ] )
			{
	#pragma unroll
				for( int r = 0; r < 16; r += 2 )
				{
					f2 e;
					if constexpr( NOARG )
					{
						// experiment ("enc_exp" 5): the argument is NOT rounded to FP16 first -- one fused multiply-add per score, s * log2 e - m * log2 e
						const float mL = mref * L2E;
						e = f2{ __builtin_amdgcn_exp2f( __builtin_fmaf( S[ r ], L2E, -mL ) ), __builtin_amdgcn_exp2f( __builtin_fmaf( S[ r + 1 ], L2E, -mL ) ) };
					}
					else
					{
						const h2 hd = __builtin_convertvector( f2{ S[ r ] - mref, S[ r + 1 ] - mref }, h2 );
						e = f2{ __builtin_amdgcn_exp2f( __builtin_fmaf( (float)hd[ 0 ], L2E, 0.0f ) ), __builtin_amdgcn_exp2f( __builtin_fmaf( (float)hd[ 1 ], L2E, 0.0f ) ) };
					}
					const h2 eh = __builtin_convertvector( e, h2 );
					P[ r >> 3 ][ r & 7 ] = eh[ 0 ];
					P[ r >> 3 ][ ( r & 7 ) + 1 ] = eh[ 1 ];
				}
			};

			float mx[ NG ];
	#pragma unroll
			for( int g = 0; g < NG; g++ ) mx[ g ] = -INFINITY;
			if constexpr( !ONLINE )
			{
				// ---- sweep 1: row maxima ----
	#pragma unroll
				for( int i = 0; i < W_NBUF - 1; i++ )
					if( i < nTiles ) issueK( i, i );
				for( int t = 0; t < nTiles; t++ )
				{
					waitTile( t, std::integral_constant<int, NG>{} );
					__syncthreads();	   // tile t visible to every wave; every wave is done with tile t - 1, whose buffer the next issue overwrites
					if( t + W_NBUF - 1 < nTiles ) issueK( t + W_NBUF - 1, ( t + W_NBUF - 1 ) % W_NBUF );
					const f16* const kt = ldsK + ( t % W_NBUF ) * F_TILE;
					auto body = [ & ]( auto last )
					{
	#pragma unroll
						for( int st = 0; st < 4; st++ )
						{
							f32x16 S[ NG ];
							scores( kt, st, t, last, S );
	#pragma unroll
							for( int g = 0; g < NG; g++ )
	#pragma unroll
								for( int r = 0; r < 16; r++ ) mx[ g ] = fmaxf( mx[ g ], S[ g ][ r ] );
						}
					};
					if( t == nTiles - 1 ) body( std::true_type{} ); else body( std::false_type{} );
				}
	#pragma unroll
				for( int g = 0; g < NG; g++ ) mx[ g ] = fmaxf( mx[ g ], __shfl_xor( mx[ g ], 32, 64 ) );
				__syncthreads();
			}

			// ---- the e / P.V sweep ----
			double sum[ NG ];
			float sumF[ NG ];
			f32x16 O[ NG ][ 2 ];
	#pragma unroll
			for( int g = 0; g < NG; g++ )
			{
				sum[ g ] = 0.0;
				sumF[ g ] = 0.0f;
	#pragma unroll
				for( int a = 0; a < 2; a++ )
	#pragma unroll
					for( int r = 0; r < 16; r++ ) O[ g ][ a ][ r ] = 0.0f;
			}
	#pragma unroll
			for( int i = 0; i < W_NBUF - 1; i++ )
				if( i < nTiles ) { issueK( i, i ); issueV( i, i ); }
			for( int t = 0; t < nTiles; t++ )
			{
				waitTile( t, std::integral_constant<int, 2 * NG>{} );
				__syncthreads();
				if( t + W_NBUF - 1 < nTiles )
				{
					issueK( t + W_NBUF - 1, ( t + W_NBUF - 1 ) % W_NBUF );
					issueV( t + W_NBUF - 1, ( t + W_NBUF - 1 ) % W_NBUF );
				}
				const f16* const kt = ldsK + ( t % W_NBUF ) * F_TILE;
				const f16* const vt = ldsV + ( t % W_NBUF ) * F_TILE;
				auto body = [ & ]( auto last )
				{
	#pragma unroll
					for( int st = 0; st < 4; st++ )
					{
						f32x16 S[ NG ];
						scores( kt, st, t, last, S );
						if constexpr( ONLINE )
						{
							float tm[ NG ];
							bool raise = false;
	#pragma unroll
							for( int g = 0; g < NG; g++ )
							{
								tm[ g ] = S[ g ][ 0 ];
	#pragma unroll
								for( int r = 1; r < 16; r++ ) tm[ g ] = fmaxf( tm[ g ], S[ g ][ r ] );
								tm[ g ] = fmaxf( tm[ g ], __shfl_xor( tm[ g ], 32, 64 ) );
								raise = raise || tm[ g ] > mx[ g ] + W_LAZY;
							}
							if( __any( raise ) )
							{
	#pragma unroll
								for( int g = 0; g < NG; g++ )
								{
									const float mNew = tm[ g ] > mx[ g ] + W_LAZY ? tm[ g ] : mx[ g ];
									const float f = __builtin_amdgcn_exp2f( ( mx[ g ] - mNew ) * L2E );	   // 1 where the maximum stays; 0 on the first sub-tile (m = -inf)
									mx[ g ] = mNew;
									sumF[ g ] *= f;
	#pragma unroll
									for( int a = 0; a < 2; a++ )
	#pragma unroll
										for( int r = 0; r < 16; r++ ) O[ g ][ a ][ r ] *= f;
								}
							}
						}
						f16x8 P[ NG ][ 2 ];
	#pragma unroll
						for( int g = 0; g < NG; g++ )
						{
							expTile( S[ g ], mx[ g ], P[ g ] );
							if constexpr( ONLINE ) sumF[ g ] += sumP( P[ g ] );
							else sum[ g ] += (double)sumP( P[ g ] );
						}
	#pragma unroll
						for( int half = 0; half < 2; half++ )
						{
							const int kb = st * 2 + half;
	#pragma unroll
							for( int ddt = 0; ddt < 2; ddt++ )
							{
								const f16x8 vf = *(const f16x8*)( vt + ( ( kb * 2 + ddt ) * 64 + lane ) * 8 );
	#pragma unroll
								for( int g = 0; g < NG; g++ ) O[ g ][ ddt ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( vf, P[ g ][ half ], O[ g ][ ddt ], 0, 0, 0 );
							}
						}
					}
				};
				if( t == nTiles - 1 ) body( std::true_type{} ); else body( std::false_type{} );
			}
	#pragma unroll
			for( int g = 0; g < NG; g++ )
			{
				if constexpr( ONLINE ) sum[ g ] = (double)sumF[ g ];
				sum[ g ] += __shfl_xor( sum[ g ], 32, 64 );
				const float invSum = (float)( 1.0 / sum[ g ] );
				if( qRow[ g ] < T )
				{
					const int b = bh / heads, h = bh - b * heads;
					f16* const o = out + ( (long long)b * T + qRow[ g ] ) * ( heads * HEAD_DIM ) + h * HEAD_DIM;
	#pragma unroll
					for( int ddt = 0; ddt < 2; ddt++ )
	#pragma unroll
						for( int g4 = 0; g4 < 4; g4++ )
						{
							f16x4 pk;
	#pragma unroll
							for( int e = 0; e < 4; e++ ) pk[ e ] = (f16)( O[ g ][ ddt ][ 4 * g4 + e ] * invSum );
							*(f16x4*)( o + ddt * 32 + 8 * g4 + 4 * hi ) = pk;
						}
				}
			}
		}

		template<int NG, bool ONLINE, bool NOARG = false>
		int launchEncWideT( const f16* q, const f16* k, const f16* vT, f16* out, int batch, int heads, int T, int Tpad, hipStream_t stream )
		{
			static PerDeviceOnce once;
			if( const int onceDev = once.needed(); onceDev >= 0 )
			{
				WH_HIP( hipFuncSetAttribute( (const void*)attentionEncW<NG, ONLINE, NOARG>, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES ) );
				once.mark( onceDev );
			}
			const int nQ = ( T + TQ - 1 ) / TQ, BH = batch * heads;
			const int xcdRemap = ( BH % 8 ) == 0 && ( g_tuning & TUNE_ATTN_XCD ) ? 1 : 0;
			hipLaunchKernelGGL( ( attentionEncW<NG, ONLINE, NOARG> ), dim3( nQ * BH ), dim3( 1024 / NG ), W_LDS_BYTES, stream, q, k, vT, out, heads, T, Tpad, nQ, xcdRemap );
			WH_HIP( hipGetLastError() );
			return 0;
		}
		// mode (the "enc_exp" option): 2 = two sweeps, 3 = one sweep
		int launchEncWide( const f16* q, const f16* k, const f16* vT, f16* out, int batch, int heads, int T, int Tpad, int mode, hipStream_t stream )
		{
			if( mode == 5 ) return launchEncWideT<2, true, true>( q, k, vT, out, batch, heads, T, Tpad, stream );
			if( mode == 3 ) return launchEncWideT<2, true>( q, k, vT, out, batch, heads, T, Tpad, stream );
			return launchEncWideT<2, false>( q, k, vT, out, batch, heads, T, Tpad, stream );
		}

		int launchEncTable( const f16* q, const f16* k, const f16* vT, f16* out, int batch, int heads, int T, int Tpad, const f16* expTab, hipStream_t stream )
		{
			static PerDeviceOnce once;
			if( const int onceDev = once.needed(); onceDev >= 0 )
			{
				WH_HIP( hipFuncSetAttribute( (const void*)attentionEncT<0>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES ) );
				WH_HIP( hipFuncSetAttribute( (const void*)attentionEncT<1>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES - T_TABLE_HALFS * 2 ) );
				once.mark( onceDev );
			}
			const int nQ = ( T + TQ - 1 ) / TQ, BH = batch * heads;
			const int xcdRemap = ( BH % 8 ) == 0 && ( g_tuning & TUNE_ATTN_XCD ) ? 1 : 0;
			if( g_opt.encExp == 2 || g_opt.encExp == 3 || g_opt.encExp == 5 ) return launchEncWide( q, k, vT, out, batch, heads, T, Tpad, g_opt.encExp, stream );
			if( g_opt.encExp == 1 )
				hipLaunchKernelGGL( attentionEncT<1>, dim3( nQ * BH ), dim3( 1024 ), T_LDS_BYTES - T_TABLE_HALFS * 2, stream, q, k, vT, out, expTab, heads, T, Tpad, nQ, xcdRemap );
			else
			hipLaunchKernelGGL( attentionEncT<0>, dim3( nQ * BH ), dim3( 1024 ), T_LDS_BYTES, stream, q, k, vT, out, expTab, heads, T, Tpad, nQ, xcdRemap );
			WH_HIP( hipGetLastError() );
			return 0;
		}

		int launchEncF( const f16* q, const f16* k, const f16* vT, f16* out, int batch, int heads, int T, int Tpad, bool exactP, hipStream_t stream )
		{
			static PerDeviceOnce once;
			if( const int onceDev = once.needed(); onceDev >= 0 )
			{
				WH_HIP( hipFuncSetAttribute( (const void*)attentionEncF<false>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BYTES ) );
				WH_HIP( hipFuncSetAttribute( (const void*)attentionEncF<true>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BYTES ) );
				once.mark( onceDev );
			}
			const int nQ = ( T + FQ - 1 ) / FQ, BH = batch * heads;
			const int xcdRemap = ( BH % 8 ) == 0 && ( g_tuning & TUNE_ATTN_XCD ) ? 1 : 0;
			const bool two = !exactP && ( g_tuning & TUNE_ATTN_ENC_2SWEEP );
			if( two )
				hipLaunchKernelGGL( attentionEncF<true>, dim3( nQ * BH ), dim3( 512 ), F_LDS_BYTES, stream, q, k, vT, out, heads, T, Tpad, nQ, xcdRemap );
			else
				hipLaunchKernelGGL( attentionEncF<false>, dim3( nQ * BH ), dim3( 512 ), F_LDS_BYTES, stream, q, k, vT, out, heads, T, Tpad, nQ, xcdRemap );
			WH_HIP( hipGetLastError() );
			return 0;
		}

		template<int KT>
		int launchEncT( const f16* q, const f16* k, const f16* vT, f16* out, int batch, int heads, int T, int Tpad, hipStream_t stream )
		{
			static PerDeviceOnce once;
			if( const int onceDev = once.needed(); onceDev >= 0 )
			{
				WH_HIP( hipFuncSetAttribute( (const void*)attentionEnc<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS_BYTES ) );
				once.mark( onceDev );
			}
			const int nQ = ( T + AQ - 1 ) / AQ, BH = batch * heads;
			const int xcdRemap = ( BH % 8 ) == 0 && ( g_tuning & TUNE_ATTN_XCD ) ? 1 : 0;
			hipLaunchKernelGGL( attentionEnc<KT>, dim3( nQ * BH ), dim3( 512 ), ATT_LDS_BYTES, stream, q, k, vT, out, heads, T, Tpad, nQ, xcdRemap );
			WH_HIP( hipGetLastError() );
			return 0;
		}
	}	// namespace

	int attentionInit() { return 0; }

	int launchAttentionEnc( const f16* q, const f16* k, const f16* vT, f16* out, int batch, int heads, int T, int Tpad, bool exactP, const f16* expTab,
		hipStream_t stream )
	{
		if( T <= 0 || T > 1536 || Tpad < ( ( T + 255 ) / 256 ) * 256 || ( Tpad & 7 ) != 0 )
		{
			setError( "attentionEnc: need 0 < T <= 1536 and Tpad >= roundup(T, 256)" );
			return -1;
		}
		// the measured path (e unnormalised in FP16, O scaled by 1 / sum) with the exponential as a table lookup; the parity flag keeps
		// the three-sweep kernel and the reference's fp16( e / sum ) operand
		// (its 1024-thread workgroups take 512 query rows each: a single window is 48 of them on 256 CUs, where attentionEncF's 96 half-size
		// workgroups finish sooner -- the table kernel runs once its grid covers the chip)
		if( expTab && !exactP && ( g_tuning & TUNE_ATTN_ENC_TABLE ) && ( g_tuning & TUNE_ATTN_ENC_F ) && ( g_tuning & TUNE_ATTN_ENC_2SWEEP ) &&
			( ( ( T + TQ - 1 ) / TQ ) * batch * heads >= 256 || ( g_tuning & TUNE_ATTN_ENC_TABLE_ANY ) ) )
			return launchEncTable( q, k, vT, out, batch, heads, T, Tpad, expTab, stream );
		if( g_tuning & TUNE_ATTN_ENC_F ) return launchEncF( q, k, vT, out, batch, heads, T, Tpad, exactP, stream );
		switch( ( T + 255 ) / 256 )
		{
		case 1: return launchEncT<1>( q, k, vT, out, batch, heads, T, Tpad, stream );
		case 2: return launchEncT<2>( q, k, vT, out, batch, heads, T, Tpad, stream );
		case 3: return launchEncT<3>( q, k, vT, out, batch, heads, T, Tpad, stream );
		case 4: return launchEncT<4>( q, k, vT, out, batch, heads, T, Tpad, stream );
		case 5: return launchEncT<5>( q, k, vT, out, batch, heads, T, Tpad, stream );
		default: return launchEncT<6>( q, k, vT, out, batch, heads, T, Tpad, stream );
		}
	}
}
