// NT GEMM kernels for gfx950: FP16 x FP16 -> FP32 on v_mfma_f32_32x32x16_f16, fused epilogues.
//
// Replaces ComputeShaders/mulMatTiled.hlsl (32x32 LDS tiles, FP32 FMA) and mulMatByRowTiled.hlsl (GEMV) of the
// reference, with the numerics of the reference's CPU path: activations are FP16 (rounded by the producer, which is
// what ggml does before every weight product, Whisper/source/ggml.c:4588-4611), weights FP16, accumulation FP32.
//
// gemmTiled<EPI, TileCfg>: every wave owns a 64x64 output sub-tile (2x2 MFMA tiles); a workgroup is 128x128x32 (4 waves,
//   3 workgroups per CU) or, for GEMMs several clips deep, 256x256x64 (16 waves). Tiles go global -> LDS directly
//   (global_load_lds_dwordx4, double buffered, one barrier per K step): the LDS image of such a load is lane-linear, so
//   rows are unpadded and the conflict-free placement is an XOR of the 16-byte chunk index with the row, applied to the
//   per-lane SOURCE address and again when the 32x32x16 fragments are read (lane l reads row l&31, chunk (l>>5)). The
//   register-staged pipeline with padded rows (144 B / 80 B, conflict free as well) is kept as the A/B alternative.
//   Block ids are remapped so that each XCD (block id % 8) owns a contiguous band of M tiles: the band's A rows are
//   fetched from HBM once per XCD and stay in that XCD's 4 MiB L2 while the (small) weight matrix is re-read from L2.
//   The epilogue requests everything it reads before its first store and does the per-row index math once per row.
// gemmSkinny: M <= 32 rows when K is not a multiple of 128. The weight matrix is the MFMA A operand (32 rows per
//   workgroup), the few activation rows are the B operand; 4 waves split K and reduce through LDS.
// gemvFused: the decode-step kernel, up to 32 activation rows (see below).
#include "kernels.h"
#include "epilogue.h"
#include <type_traits>

namespace wh
{
	namespace
	{

		// Tile configuration: every wave owns a 64x64 sub-tile (2x2 MFMA 32x32x16 tiles), waves are laid out WAVES_M x WAVES_N.
		// MINW = waves per SIMD the register allocator must leave room for (blocks per CU * waves per block / 4).
		// GL = tiles go global -> LDS directly (global_load_lds_dwordx4, no staging registers): the LDS image of a wave's
		// instruction is lane-linear (base + lane * 16 bytes), so rows are unpadded and the bank-conflict-free placement is
		// an XOR of the 16-byte chunk index applied to the SOURCE address and again when the fragments are read.
		// A wave owns TI x TJ MFMA tiles of 32x32 (default 2 x 2 = 64x64); 4 x 2 reads 6 fragments for 8 MFMAs instead of 4 for 4,
		// which is what the LDS bandwidth of a CU asks for.
		// NBUF (GL only) = LDS stages: 2 = the next tile lands while this one is multiplied (wait for everything at the top of
		// a K step); 3 or 4 = one or two MORE tiles stay in flight across the step's barrier (counted vmcnt + raw s_barrier),
		// which is what covers an HBM round trip that is longer than one K step.
		// PIPE: see below (fragment prefetch / loads spread behind the MFMA groups).
		template<int BM_, int BN_, int BK_, int MINW_, int PF_, bool GL_ = false, int TI_ = 2, int TJ_ = 2, int NBUF_ = 2, int PIPE_ = 0>
		struct TileCfg
		{
			static constexpr int BM = BM_, BN = BN_, BK = BK_, MINW = MINW_, PF = PF_, TI = TI_, TJ = TJ_, NBUF = NBUF_;
			// PIPE (GL only): 1 = FRAGPF, the MFMA fragments of k-substep s+1 are read from LDS before the MFMAs of substep s are
			// issued (two register sets; hipcc on its own re-uses one set, so every substep starts with an exposed LDS round trip).
			// (Issuing the next tile's direct-to-LDS loads one or two at a time behind the MFMA groups instead of as a burst at the
			// top of the K step was measured too: no difference, profiles/r02_gemm_kloop_ablation.txt.)
			static constexpr bool FRAGPF = PIPE_ >= 1;
			static constexpr bool GL = GL_;
			static constexpr int WAVES_M = BM / ( 32 * TI ), WAVES_N = BN / ( 32 * TJ ), NT = WAVES_M * WAVES_N * 64;
			static_assert( GL || ( TI == 2 && TJ == 2 ), "the register-staged path is written for 64x64 wave tiles" );
			static constexpr int STRIDE = GL ? BK : BK + 8;		 // halfs per LDS row: padded 144 B (BK 64) / 80 B (BK 32) are conflict free
			static constexpr int RPI = 512 / BK;				 // GL: tile rows one wave instruction covers (1 KB)
			static constexpr int RPB = 128 / BK;				 // GL: tile rows per 256-byte bank row
			static constexpr int IA = BM / RPI / ( NT / 64 ), IW = BN / RPI / ( NT / 64 );	 // GL: instructions per wave and tile
			static constexpr int A_HALFS = BM * STRIDE, W_HALFS = BN * STRIDE, STAGE = A_HALFS + W_HALFS;
			// the LDS-transposed epilogue (tileEpilogueWide) takes 8 KiB per wave once the operand tiles are dead
			static constexpr int LDS_BYTES = ( NBUF * STAGE * 2 > ( GL ? NT / 64 * 8192 : 0 ) ) ? NBUF * STAGE * 2 : NT / 64 * 8192;
			static_assert( NBUF == 2 || GL, "more than two stages only with direct-to-LDS staging" );
			static constexpr int CPR = BK / 8;					 // 16-byte chunks per tile row
			static constexpr int CA = BM * CPR / NT, CW = BN * CPR / NT;
			static_assert( CA >= 1 && CW >= 1 && BM * CPR % NT == 0 && BN * CPR % NT == 0, "tile does not divide over the threads" );
		};
		// Measured on MI355X (tools/gemm_probe.py, profiles/r01_gemm_tile_probe.txt), M = 10500: 256x256x64 wins when the grid
		// still fills the chip (N >= 2048: 593-662 TFLOP/s), 128x128x32 (3 blocks per CU) wins on narrow outputs and small M;
		// the two-tile-deep prefetch (PF = 2) measured 3-5 % slower than PF = 1 at every shape.
		using CfgDefault = TileCfg<128, 128, 32, 3, 1>;
		using CfgBig = TileCfg<256, 256, 64, 4, 1>;
		using CfgGl = TileCfg<128, 128, 32, 3, 1, true>;
		using CfgGlBig = TileCfg<256, 256, 64, 4, 1, true>;
		using CfgGlPf = TileCfg<128, 128, 32, 3, 1, true, 2, 2, 2, 1>;
		using CfgGlBigPf = TileCfg<256, 256, 64, 4, 1, true, 2, 2, 2, 1>;

		// one 16-byte-per-lane global -> LDS instruction; M0 (the LDS destination base) is saved and restored inside the
		// statement because the compiler does not preserve it around inline assembly (cdna_hip_programming.md section 5.7)
		__device__ __forceinline__ void ldsDma16( const void* src, unsigned ldsByteAddr )
		{
			unsigned keep;
			asm volatile( "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
						  : "=&s"( keep )
						  : "v"( src ), "s"( ldsByteAddr )
						  : "memory" );
		}
		// physical position (in halfs) of logical 16-byte chunk c of tile row `row` in a GL tile
		template<class C>
		__device__ __forceinline__ int glOffset( int row, int c )
		{
			return row * C::BK + ( ( c ^ ( ( row / C::RPB ) % C::CPR ) ) << 3 );
		}

		// Tile epilogue shared by the staging variants: D[row][col], col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
		// Same arithmetic per element as epilogueOne, organised for the memory system: the row-dependent index math (the
		// divisions by T) is done once per row instead of once per element, and everything the epilogue READS (residual,
		// position embedding, bias) is requested -- with clamped, hence unconditional, addresses -- before the first store, so
		// a wave pays one memory round trip instead of one per element (the residual is updated in place: a load may not be
		// moved above the preceding store by the compiler).
		template<int EPI, class C>
		__device__ __forceinline__ void tileEpilogue( const GemmArgs& a, f32x16 ( &acc )[ C::TI ][ C::TJ ], int tm, int tn, int wm, int wn, int lane )
		{
			constexpr int BM = C::BM, BN = C::BN;
			const int hi = lane >> 5;
			const int d = a.H * HEAD_DIM;
			int nn[ C::TJ ];
			float bias[ C::TJ ];
#pragma unroll
			for( int j = 0; j < C::TJ; j++ )
			{
				nn[ j ] = tn * BN + wn * 32 * C::TJ + j * 32 + ( lane & 31 );
				const int nc = nn[ j ] < a.N ? nn[ j ] : a.N - 1;
				bias[ j ] = a.bias ? a.bias[ nc ] : 0.0f;
			}
#pragma unroll
			for( int i = 0; i < C::TI; i++ )
			{
				const int mBase = tm * BM + wm * 32 * C::TI + i * 32 + 4 * hi;
				if constexpr( EPI == EPI_F32 || EPI == EPI_CONV2 )
				{
					long long ro[ 16 ], po[ 16 ];
#pragma unroll
					for( int r = 0; r < 16; r++ )
					{
						int m = mBase + ( r & 3 ) + 8 * ( r >> 2 );
						m = m < a.M ? m : a.M - 1;
						if constexpr( EPI == EPI_F32 )
							ro[ r ] = rowOffset( m, a.Mb, a.ldc, a.cBatchStride );
						else
						{
							const int b = m / a.Mb;
							ro[ r ] = (long long)m * a.ldc;
							po[ r ] = (long long)( m - b * a.Mb ) * a.N;
						}
					}
					float ex[ C::TJ ][ 16 ];
#pragma unroll
					for( int j = 0; j < C::TJ; j++ )
					{
						const int nc = nn[ j ] < a.N ? nn[ j ] : a.N - 1;
#pragma unroll
						for( int r = 0; r < 16; r++ )
						{
							if constexpr( EPI == EPI_F32 )
								ex[ j ][ r ] = a.res ? a.res[ ro[ r ] + nc ] : 0.0f;
							else
								ex[ j ][ r ] = a.pe[ po[ r ] + nc ];
						}
					}
#pragma unroll
					for( int j = 0; j < C::TJ; j++ )
					{
						if( nn[ j ] >= a.N ) continue;
#pragma unroll
						for( int r = 0; r < 16; r++ )
						{
							const int m = mBase + ( r & 3 ) + 8 * ( r >> 2 );
							if( m >= a.M ) continue;
							if constexpr( EPI == EPI_F32 )
								a.out32[ ro[ r ] + nn[ j ] ] = ( acc[ i ][ j ][ r ] + bias[ j ] ) + ex[ j ][ r ];
							else
								a.out32[ ro[ r ] + nn[ j ] ] = ex[ j ][ r ] + (float)gelu16( acc[ i ][ j ][ r ] + bias[ j ] );
						}
					}
				}
				else if constexpr( EPI == EPI_F16_GELU )
				{
#pragma unroll
					for( int r = 0; r < 16; r++ )
					{
						const int m = mBase + ( r & 3 ) + 8 * ( r >> 2 );
						if( m >= a.M ) continue;
						const long long ro = rowOffset( m, a.Mb, a.ldc, a.cBatchStride );
#pragma unroll
						for( int j = 0; j < C::TJ; j++ )
							if( nn[ j ] < a.N ) a.out16[ ro + nn[ j ] ] = gelu16( acc[ i ][ j ][ r ] + bias[ j ] );
					}
				}
				else if constexpr( EPI == EPI_QKV_ENC || EPI == EPI_CROSS_KV )
				{
					// column-dependent part of the destination, once per j
					int sel[ C::TJ ];
					long long colOff[ C::TJ ];
#pragma unroll
					for( int j = 0; j < C::TJ; j++ )
					{
						const int n = nn[ j ] < a.N ? nn[ j ] : a.N - 1;
						if constexpr( EPI == EPI_QKV_ENC )
						{
							sel[ j ] = n / d;
							const int c = n - sel[ j ] * d;
							colOff[ j ] = (long long)( c >> 6 ) * ( sel[ j ] == 2 ? (long long)HEAD_DIM * a.Tpad : (long long)a.T * HEAD_DIM ) + ( sel[ j ] == 2 ? 0 : ( c & 63 ) );
						}
						else
						{
							const int layer = n / ( 2 * d );
							const int c2 = n - layer * 2 * d;
							sel[ j ] = c2 >= d ? 1 : 0;
							const int c = sel[ j ] ? c2 - d : c2;
							colOff[ j ] = ( (long long)layer * a.B * a.H + ( c >> 6 ) ) * a.T * HEAD_DIM + ( c & 63 );
						}
					}
					const bool packT = ( a.T & 3 ) == 0;
#pragma unroll
					for( int g = 0; g < 4; g++ )
					{
						// rows mBase + 8 g + {0,1,2,3}: 4 consecutive time steps of one sequence when T % 4 == 0
						const int m0 = mBase + 8 * g;
						const int mc = m0 < a.M ? m0 : a.M - 1;
						const int b0 = mc / a.T;
						const int t0 = mc - b0 * a.T;
#pragma unroll
						for( int j = 0; j < C::TJ; j++ )
						{
							if( nn[ j ] >= a.N ) continue;
							if constexpr( EPI == EPI_QKV_ENC )
							{
								if( sel[ j ] == 2 && packT )
								{
									// fragment-major V: the 4 rows are 4 consecutive keys = 4 consecutive halfs of one fragment
									if( m0 < a.M )
									{
										const int c = nn[ j ] - 2 * d;
										f16x4 pk;
#pragma unroll
										for( int e = 0; e < 4; e++ ) pk[ e ] = (f16)( acc[ i ][ j ][ 4 * g + e ] + bias[ j ] );
										*(f16x4*)( a.v + (long long)b0 * a.H * HEAD_DIM * a.Tpad + colOff[ j ] + vFragIndex( t0, c & 63 ) ) = pk;
									}
									continue;
								}
							}
#pragma unroll
							for( int e = 0; e < 4; e++ )
							{
								const int m = m0 + e;
								if( m >= a.M ) continue;
								int b = b0, t = t0 + e;
								if( !packT && t >= a.T )
								{
									b = m / a.T;
									t = m - b * a.T;
								}
								const float v = acc[ i ][ j ][ 4 * g + e ];
								if constexpr( EPI == EPI_QKV_ENC )
								{
									const float x = v + bias[ j ];
									if( sel[ j ] == 0 )
										a.q[ ( (long long)b * a.H * a.T + t ) * HEAD_DIM + colOff[ j ] ] = (f16)x;
									else if( sel[ j ] == 1 )
										a.k[ ( (long long)b * a.H * a.T + t ) * HEAD_DIM + colOff[ j ] ] = (f16)x;
									else
										a.v[ (long long)b * a.H * HEAD_DIM * a.Tpad + colOff[ j ] + vFragIndex( t, ( nn[ j ] - 2 * d ) & 63 ) ] = (f16)x;
								}
								else
								{
									const long long o = ( (long long)b * a.H * a.T + t ) * HEAD_DIM + colOff[ j ];
									if( sel[ j ] )
										a.v[ o ] = (f16)( v + bias[ j ] );
									else
										a.k[ o ] = (f16)( v * a.scale );
								}
							}
						}
					}
				}
				else
				{
#pragma unroll
					for( int j = 0; j < C::TJ; j++ )
					{
						if( nn[ j ] >= a.N ) continue;
#pragma unroll
						for( int r = 0; r < 16; r++ )
						{
							const int m = mBase + ( r & 3 ) + 8 * ( r >> 2 );
							if( m < a.M )
								epilogueOne<EPI>( a, m, nn[ j ], acc[ i ][ j ][ r ] );
						}
					}
				}
			}
		}

		// ---------------------------------------------------------------------------------------------------------------
		// Wide epilogue: the wave's 64x64 accumulator block goes through the (now idle) LDS tile memory and leaves as 16-byte
		// stores along the rows of the destination. In the MFMA accumulator layout a lane holds ONE column and 16 rows of
		// each 32x32 tile, so a direct epilogue issues 64 two- or four-byte stores per lane (and as many residual loads);
		// per 256x256 tile that is 1024 wave-level store instructions of 64-128 useful bytes, and the tile's fixed cost
		// (24 us against 28 us of K loop at K = 1024, profiles/r01_gemm_tile_probe.txt) was mostly their issue time.
		// Through LDS a lane stores 8 x 16 bytes (FP16 outputs) or loads + stores 16 x 16 bytes (FP32 outputs with residual).
		// LDS image per wave: [64 rows][64 cols] FP16 (8 KiB) or [32 rows][64 cols] FP32 (8 KiB, two halves), 16-byte chunk
		// index XORed with the row so that both the column-wise writes and the row-wise reads are conflict free.
		// Same arithmetic per element as the direct epilogue. Preconditions (checked by the launcher, a.wideEpi): N % 8 == 0,
		// 16-byte aligned rows, T % 8 == 0 irrelevant (rows are independent), a wave's 64 columns inside one head.
		template<int EPI, class C>
		__device__ __forceinline__ void tileEpilogueWide( const GemmArgs& a, f32x16 ( &acc )[ C::TI ][ C::TJ ], int tm, int tn, int wm, int wn, int lane,
			unsigned char* ldsWave )
		{
			static_assert( C::TI == 2 && C::TJ == 2, "64x64 wave tiles" );
			constexpr int BM = C::BM, BN = C::BN;
			const int hi = lane >> 5, c = lane & 31;
			const int d = a.H * HEAD_DIM;
			const int m0 = tm * BM + wm * 64, n0 = tn * BN + wn * 64;
			float bias[ 2 ];
	#pragma unroll
			for( int j = 0; j < 2; j++ )
			{
				const int n = n0 + j * 32 + c;
				bias[ j ] = ( a.bias && n < a.N ) ? a.bias[ n ] : 0.0f;
			}
			if constexpr( EPI == EPI_F16_GELU || EPI == EPI_QKV_ENC || EPI == EPI_CROSS_KV )
			{
				f16* const L = (f16*)ldsWave;
				// column block -> what it is (uniform over the wave: 64 columns never straddle a head)
				int sel = 0, head = 0, layer = 0;
				if constexpr( EPI == EPI_QKV_ENC )
				{
					sel = n0 / d;
					head = ( n0 - sel * d ) >> 6;
				}
				if constexpr( EPI == EPI_CROSS_KV )
				{
					layer = n0 / ( 2 * d );
					const int c2 = n0 - layer * 2 * d;
					sel = c2 >= d ? 1 : 0;
					head = ( sel ? c2 - d : c2 ) >> 6;
				}
	#pragma unroll
				for( int i = 0; i < 2; i++ )
	#pragma unroll
					for( int j = 0; j < 2; j++ )
	#pragma unroll
						for( int r = 0; r < 16; r++ )
						{
							const int row = i * 32 + ( r & 3 ) + 8 * ( r >> 2 ) + 4 * hi;
							const int col = j * 32 + c;
							const float v = acc[ i ][ j ][ r ];
							f16 hv;
							if constexpr( EPI == EPI_F16_GELU )
								hv = gelu16( v + bias[ j ] );
							else if constexpr( EPI == EPI_QKV_ENC )
								hv = (f16)( v + bias[ j ] );
							else
								hv = sel ? (f16)( v + bias[ j ] ) : (f16)( v * a.scale );
							L[ row * 64 + ( ( ( col >> 3 ) ^ ( row & 7 ) ) << 3 ) + ( col & 7 ) ] = hv;
						}
				__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
				const int chunk = lane & 7;
	#pragma unroll
				for( int it = 0; it < 8; it++ )
				{
					const int row = it * 8 + ( lane >> 3 );
					const int m = m0 + row;
					const f16x8 v = *(const f16x8*)( L + row * 64 + ( ( chunk ^ ( row & 7 ) ) << 3 ) );
					const int n = n0 + chunk * 8;
					if( m >= a.M || n >= a.N ) continue;
					if constexpr( EPI == EPI_F16_GELU )
						*(f16x8*)( a.out16 + rowOffset( m, a.Mb, a.ldc, a.cBatchStride ) + n ) = v;
					else
					{
						const int b = m / a.T;
						const int t = m - b * a.T;
						if constexpr( EPI == EPI_QKV_ENC )
						{
							f16* const dst = sel == 0 ? a.q : a.k;
							*(f16x8*)( dst + ( ( (long long)b * a.H + head ) * a.T + t ) * HEAD_DIM + chunk * 8 ) = v;
						}
						else
						{
							f16* const dst = sel ? a.v : a.k;
							*(f16x8*)( dst + ( ( ( (long long)layer * a.B + b ) * a.H + head ) * a.T + t ) * HEAD_DIM + chunk * 8 ) = v;
						}
					}
				}
			}
			else
			{
				// FP32 outputs: 32 rows at a time
				float* const L = (float*)ldsWave;
				const int chunk = lane & 15;
	#pragma unroll
				for( int i = 0; i < 2; i++ )
				{
					if( i == 1 )
					{
						__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
						__builtin_amdgcn_wave_barrier();
						__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
					}
	#pragma unroll
					for( int j = 0; j < 2; j++ )
	#pragma unroll
						for( int r = 0; r < 16; r++ )
						{
							const int row = ( r & 3 ) + 8 * ( r >> 2 ) + 4 * hi;
							const int col = j * 32 + c;
							float v = acc[ i ][ j ][ r ] + bias[ j ];
							if constexpr( EPI == EPI_CONV2 ) v = (float)gelu16( v );
							L[ row * 64 + ( ( ( col >> 2 ) ^ ( row & 15 ) ) << 2 ) + ( col & 3 ) ] = v;
						}
					__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
					// everything a group of 4 chunks READS from memory first, then its stores (two groups per half: 16 + 8 registers
					// of operands in flight instead of 32 + 16)
	#pragma unroll
					for( int g4 = 0; g4 < 2; g4++ )
					{
						f32x4 ex[ 4 ];
						long long off[ 4 ];
	#pragma unroll
						for( int u = 0; u < 4; u++ )
						{
							const int row = ( g4 * 4 + u ) * 4 + ( lane >> 4 );
							int m = m0 + i * 32 + row;
							m = m < a.M ? m : a.M - 1;
							int n = n0 + chunk * 4;
							n = n < a.N ? n : a.N - 4;
							if constexpr( EPI == EPI_F32 )
							{
								off[ u ] = rowOffset( m, a.Mb, a.ldc, a.cBatchStride ) + n;
								ex[ u ] = a.res ? *(const f32x4*)( a.res + off[ u ] ) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
							}
							else
							{
								const int b = m / a.Mb;
								off[ u ] = (long long)m * a.ldc + n;
								ex[ u ] = *(const f32x4*)( a.pe + (long long)( m - b * a.Mb ) * a.N + n );
							}
						}
	#pragma unroll
						for( int u = 0; u < 4; u++ )
						{
							const int row = ( g4 * 4 + u ) * 4 + ( lane >> 4 );
							const int m = m0 + i * 32 + row;
							const int n = n0 + chunk * 4;
							if( m >= a.M || n >= a.N ) continue;
							const f32x4 v = *(const f32x4*)( L + row * 64 + ( ( chunk ^ ( row & 15 ) ) << 2 ) );
							f32x4 o;
	#pragma unroll
							for( int e = 0; e < 4; e++ ) o[ e ] = EPI == EPI_F32 ? v[ e ] + ex[ u ][ e ] : ex[ u ][ e ] + v[ e ];
							*(f32x4*)( a.out32 + off[ u ] ) = o;
						}
					}
				}
			}
		}

		template<int EPI, class C, bool WIDE = false>
		__global__ void __launch_bounds__( C::NT, C::MINW ) gemmTiled( const GemmArgs a )
		{
			constexpr int BM = C::BM, BN = C::BN, BK = C::BK, LDS_STRIDE = C::STRIDE;
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smem[];
			f16* const lds = (f16*)smem;

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = tid >> 6;
			const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;

			const int tilesN = ( a.N + BN - 1 ) / BN;
			// XCD-aware, bijective block remap (each XCD gets a contiguous range of linear tile ids)
			int lin;
			{
				const int nb = gridDim.x, bid = blockIdx.x;
				const int q = nb >> 3, r = nb & 7;
				const int xcd = bid & 7, idx = bid >> 3;
				lin = ( xcd < r ? xcd * ( q + 1 ) : r * ( q + 1 ) + ( xcd - r ) * q ) + idx;
			}
			// Walk order inside an XCD's range. Row-major (tm = lin / tilesN) makes the ~32 (256x256) or ~96 (128x128) tiles an
			// XCD has in flight share ONE A tile and sweep that many different W tiles through a 4 MiB L2, so W is re-read from
			// the fabric once per M tile row (measured 8.3 GB for 0.12 GB of operands on the cross-KV product,
			// profiles/r01_pmc_hbm_traffic.csv). Bands of groupM M tiles, walked column by column, keep the band's A rows
			// (groupM x BM x K halves) resident while every W tile is fetched once per band and shared by groupM tiles.
			int tm, tn;
			if( a.groupM > 1 )
			{
				const int tilesM = ( a.M + BM - 1 ) / BM;
				const int perBand = a.groupM * tilesN;
				const int band = lin / perBand;
				const int first = band * a.groupM;
				const int rows = min( tilesM - first, a.groupM );
				const int r = lin - band * perBand;
				tm = first + r % rows;
				tn = r / rows;
			}
			else
			{
				tm = lin / tilesN;
				tn = lin - tm * tilesN;
			}

			if constexpr( C::GL )
			{
				// ---- direct-to-LDS pipeline: one barrier per K step, tile kt+1 lands while tile kt is multiplied ----
				const f16* gA[ C::IA ];
				const f16* gW[ C::IW ];
				const int rIn = lane / C::CPR, cPhys = lane % C::CPR;
#pragma unroll
				for( int i = 0; i < C::IA; i++ )
				{
					const int row = ( wave * C::IA + i ) * C::RPI + rIn;
					const int c = cPhys ^ ( ( row / C::RPB ) % C::CPR );
					int m = tm * BM + row;
					m = m < a.M ? m : a.M - 1;
					gA[ i ] = a.A + rowOffset( m, a.Mb, a.lda, a.aBatchStride ) + c * 8;
				}
#pragma unroll
				for( int i = 0; i < C::IW; i++ )
				{
					const int row = ( wave * C::IW + i ) * C::RPI + rIn;
					const int c = cPhys ^ ( ( row / C::RPB ) % C::CPR );
					int n = tn * BN + row;
					n = n < a.N ? n : a.N - 1;
					gW[ i ] = a.W + (long long)n * a.K + c * 8;
				}
				f32x16 acc[ C::TI ][ C::TJ ];
#pragma unroll
				for( int i = 0; i < C::TI; i++ )
#pragma unroll
					for( int j = 0; j < C::TJ; j++ )
#pragma unroll
						for( int r = 0; r < 16; r++ )
							acc[ i ][ j ][ r ] = 0.0f;
				const int nk = a.K / BK;
				const int fragRow = lane & 31;
				const int fragC = lane >> 5;
				typedef __attribute__( ( address_space( 3 ) ) ) void* LdsPtr;
				typedef const __attribute__( ( address_space( 1 ) ) ) void* GlobalPtr;
				// the LDS-DMA instructions p0 .. p1-1 of tile kt (A pieces first, then W pieces)
				auto issuePieces = [ & ]( int kt, int buf, int p0, int p1 )
				{
					f16* const dstA = lds + buf * C::STAGE + wave * C::IA * C::RPI * BK;
					f16* const dstW = lds + buf * C::STAGE + C::A_HALFS + wave * C::IW * C::RPI * BK;
					const int ko = kt * BK;
					if constexpr( C::FRAGPF )
					{
						// Issued as assembly: hipcc models the builtin as a FLAT access that may touch LDS and, while one is in
						// flight, turns every LDS wait of the wave into lgkmcnt(0) -- the fragment prefetch below needs counted
						// waits. The loads are ordered by the explicit vmcnt waits + barriers of the K loop.
						const unsigned baseA = __builtin_amdgcn_readfirstlane( (unsigned)(size_t)(LdsPtr)dstA );
						const unsigned baseW = __builtin_amdgcn_readfirstlane( (unsigned)(size_t)(LdsPtr)dstW );
	#pragma unroll
						for( int i = 0; i < C::IA; i++ )
							if( i >= p0 && i < p1 )
								ldsDma16( gA[ i ] + ko, baseA + i * C::RPI * BK * 2 );
	#pragma unroll
						for( int i = 0; i < C::IW; i++ )
							if( C::IA + i >= p0 && C::IA + i < p1 )
								ldsDma16( gW[ i ] + ko, baseW + i * C::RPI * BK * 2 );
						return;
					}
#pragma unroll
					for( int i = 0; i < C::IA; i++ )
						__builtin_amdgcn_global_load_lds( (GlobalPtr)( gA[ i ] + ko ), (LdsPtr)( dstA + i * C::RPI * BK ), 16, 0, 0 );
#pragma unroll
					for( int i = 0; i < C::IW; i++ )
						__builtin_amdgcn_global_load_lds( (GlobalPtr)( gW[ i ] + ko ), (LdsPtr)( dstW + i * C::RPI * BK ), 16, 0, 0 );
				};
				constexpr int NB = C::NBUF;
				constexpr int PER_TILE = C::IA + C::IW;	  // LDS-DMA instructions of one tile per wave
				auto issue = [ & ]( int kt, int buf ) { issuePieces( kt, buf, 0, PER_TILE ); };
	#pragma unroll
				for( int p = 0; p < NB - 1; p++ )
					if( p < nk ) issue( p, p );
				for( int kt = 0; kt < nk; kt++ )
				{
					const int buf = kt % NB;
					if constexpr( NB == 2 )
					{
						asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
						__syncthreads();
					}
					else
					{
						// tile kt must have landed; the NB - 2 tiles behind it may stay in flight (they were issued later and
						// complete in order). A plain __syncthreads() would drain them: raw barrier.
						if( kt + NB - 2 < nk )
							asm volatile( "s_waitcnt vmcnt(%0)" ::"n"( ( NB - 2 ) * PER_TILE ) : "memory" );
						else
							asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
						__builtin_amdgcn_s_barrier();
					}
					if( kt + NB - 1 < nk ) issue( kt + NB - 1, ( kt + NB - 1 ) % NB );
					const f16* const ldsA = lds + buf * C::STAGE;
					const f16* const ldsW = ldsA + C::A_HALFS;
					if constexpr( C::FRAGPF )
					{
						f16x8 fa[ 2 ][ C::TI ], fb[ 2 ][ C::TJ ];
						auto readFrags = [ & ]( auto set, int ks )
						{
							constexpr int S = decltype( set )::value;
	#pragma unroll
							for( int i = 0; i < C::TI; i++ )
								fa[ S ][ i ] = *(const f16x8*)( ldsA + glOffset<C>( wm * 32 * C::TI + i * 32 + fragRow, ks * 2 + fragC ) );
	#pragma unroll
							for( int j = 0; j < C::TJ; j++ )
								fb[ S ][ j ] = *(const f16x8*)( ldsW + glOffset<C>( wn * 32 * C::TJ + j * 32 + fragRow, ks * 2 + fragC ) );
						};
						auto mfmas = [ & ]( auto set )
						{
							constexpr int S = decltype( set )::value;
	#pragma unroll
							for( int i = 0; i < C::TI; i++ )
	#pragma unroll
								for( int j = 0; j < C::TJ; j++ )
									acc[ i ][ j ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( fa[ S ][ i ], fb[ S ][ j ], acc[ i ][ j ], 0, 0, 0 );
						};
						using S0 = std::integral_constant<int, 0>;
						using S1 = std::integral_constant<int, 1>;
						static_assert( ( BK / 16 ) % 2 == 0, "fragment prefetch walks the k-substeps in pairs" );
						readFrags( S0{}, 0 );
	#pragma unroll
						for( int ks = 0; ks < BK / 16; ks += 2 )
						{
							// the scheduling fences keep hipcc from sinking the reads back below the MFMAs to save registers
							readFrags( S1{}, ks + 1 );
							__builtin_amdgcn_sched_barrier( 0 );
							mfmas( S0{} );
							__builtin_amdgcn_sched_barrier( 0 );
							if( ks + 2 < BK / 16 ) readFrags( S0{}, ks + 2 );
							__builtin_amdgcn_sched_barrier( 0 );
							mfmas( S1{} );
							__builtin_amdgcn_sched_barrier( 0 );
						}
					}
					else
					{
	#pragma unroll
					for( int ks = 0; ks < BK / 16; ks++ )
					{
						f16x8 fa[ C::TI ], fb[ C::TJ ];
	#pragma unroll
						for( int i = 0; i < C::TI; i++ )
							fa[ i ] = *(const f16x8*)( ldsA + glOffset<C>( wm * 32 * C::TI + i * 32 + fragRow, ks * 2 + fragC ) );
	#pragma unroll
						for( int j = 0; j < C::TJ; j++ )
							fb[ j ] = *(const f16x8*)( ldsW + glOffset<C>( wn * 32 * C::TJ + j * 32 + fragRow, ks * 2 + fragC ) );
	#pragma unroll
						for( int i = 0; i < C::TI; i++ )
	#pragma unroll
							for( int j = 0; j < C::TJ; j++ )
								acc[ i ][ j ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( fa[ i ], fb[ j ], acc[ i ][ j ], 0, 0, 0 );
					}
					}
				}
				if constexpr( WIDE )
				{
					// V of the encoder (fragment-major, already 8-byte stores of 4 keys) keeps the direct path; a wave's 64 columns are one head
					const bool vPart = EPI == EPI_QKV_ENC && ( tn * BN + wn * 64 ) >= 2 * a.H * HEAD_DIM;
					__syncthreads();	  // every wave is done reading the operand tiles: LDS is free
					if( !vPart )
					{
						tileEpilogueWide<EPI, C>( a, acc, tm, tn, wm, wn, lane, smem + wave * 8192 );
						return;
					}
				}
				tileEpilogue<EPI, C>( a, acc, tm, tn, wm, wn, lane );
			}
			else
			{
			// global -> register staging: CA / CW chunks of 16 bytes per thread
			const f16* gA[ C::CA ];
			const f16* gW[ C::CW ];
			int offA[ C::CA ], offW[ C::CW ];
#pragma unroll
			for( int i = 0; i < C::CA; i++ )
			{
				const int c = tid + i * C::NT;
				const int row = c / C::CPR;
				const int kc = ( c % C::CPR ) * 8;
				int m = tm * BM + row;
				m = m < a.M ? m : a.M - 1;
				gA[ i ] = a.A + rowOffset( m, a.Mb, a.lda, a.aBatchStride ) + kc;
				offA[ i ] = row * LDS_STRIDE + kc;
			}
#pragma unroll
			for( int i = 0; i < C::CW; i++ )
			{
				const int c = tid + i * C::NT;
				const int row = c / C::CPR;
				const int kc = ( c % C::CPR ) * 8;
				int n = tn * BN + row;
				n = n < a.N ? n : a.N - 1;
				gW[ i ] = a.W + (long long)n * a.K + kc;
				offW[ i ] = C::A_HALFS + row * LDS_STRIDE + kc;
			}

			f32x16 acc[ 2 ][ 2 ];
#pragma unroll
			for( int i = 0; i < 2; i++ )
#pragma unroll
				for( int j = 0; j < 2; j++ )
#pragma unroll
					for( int r = 0; r < 16; r++ )
						acc[ i ][ j ][ r ] = 0.0f;

			// Register prefetch, PF tiles deep: while tile kt is consumed from LDS, tile kt+1 sits in a register set (written
			// to the other LDS buffer after the MFMAs) and, with PF == 2, the loads of tile kt+2 are already in flight in the
			// second set. The loop is unrolled by two so that the sets are statically indexed.
			u32x4 ra[ 2 ][ C::CA ], rw[ 2 ][ C::CW ];
			const int nk = a.K / BK;
			const int fragRow = lane & 31;
			const int fragK = ( lane >> 5 ) * 8;

			auto loadTile = [ & ]( auto set, int kt )
			{
				constexpr int S = decltype( set )::value;
				const int ko = kt * BK;
#pragma unroll
				for( int i = 0; i < C::CA; i++ ) ra[ S ][ i ] = *(const u32x4*)( gA[ i ] + ko );
#pragma unroll
				for( int i = 0; i < C::CW; i++ ) rw[ S ][ i ] = *(const u32x4*)( gW[ i ] + ko );
			};
			auto storeTile = [ & ]( auto set, int buf )
			{
				constexpr int S = decltype( set )::value;
				f16* const dst = lds + buf * C::STAGE;
#pragma unroll
				for( int i = 0; i < C::CA; i++ ) *(u32x4*)( dst + offA[ i ] ) = ra[ S ][ i ];
#pragma unroll
				for( int i = 0; i < C::CW; i++ ) *(u32x4*)( dst + offW[ i ] ) = rw[ S ][ i ];
			};
			auto compute = [ & ]( int buf )
			{
				const f16* const ldsA = lds + buf * C::STAGE;
				const f16* const ldsW = ldsA + C::A_HALFS;
#pragma unroll
				for( int ks = 0; ks < BK / 16; ks++ )
				{
					f16x8 fa[ 2 ], fb[ 2 ];
#pragma unroll
					for( int i = 0; i < 2; i++ )
					{
						fa[ i ] = *(const f16x8*)( ldsA + ( wm * 64 + i * 32 + fragRow ) * LDS_STRIDE + ks * 16 + fragK );
						fb[ i ] = *(const f16x8*)( ldsW + ( wn * 64 + i * 32 + fragRow ) * LDS_STRIDE + ks * 16 + fragK );
					}
#pragma unroll
					for( int i = 0; i < 2; i++ )
#pragma unroll
						for( int j = 0; j < 2; j++ )
							acc[ i ][ j ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( fa[ i ], fb[ j ], acc[ i ][ j ], 0, 0, 0 );
				}
			};
			using Set0 = std::integral_constant<int, 0>;
			using Set1 = std::integral_constant<int, 1>;

			if constexpr( C::PF == 2 )
			{
				loadTile( Set0{}, 0 );
				if( nk > 1 ) loadTile( Set1{}, 1 );
				storeTile( Set0{}, 0 );
				__syncthreads();
				for( int kt = 0; kt < nk; kt += 2 )
				{
					// even step: tile kt in LDS buffer 0, tile kt+1 in register set 1
					if( kt + 2 < nk ) loadTile( Set0{}, kt + 2 );
					compute( 0 );
					if( kt + 1 < nk ) storeTile( Set1{}, 1 );
					__syncthreads();
					if( kt + 1 >= nk ) break;
					// odd step: tile kt+1 in LDS buffer 1, tile kt+2 in register set 0
					if( kt + 3 < nk ) loadTile( Set1{}, kt + 3 );
					compute( 1 );
					if( kt + 2 < nk ) storeTile( Set0{}, 0 );
					__syncthreads();
				}
			}
			else
			{
				loadTile( Set0{}, 0 );
				storeTile( Set0{}, 0 );
				__syncthreads();
				for( int kt = 0; kt < nk; kt++ )
				{
					const int cur = kt & 1;
					if( kt + 1 < nk ) loadTile( Set0{}, kt + 1 );
					compute( cur );
					if( kt + 1 < nk ) storeTile( Set0{}, cur ^ 1 );
					__syncthreads();
				}
			}

			tileEpilogue<EPI, C>( a, acc, tm, tn, wm, wn, lane );
			}
		}

		// ---------------------------------------------------------------------------------------------------------------
		// gemmTiled8: the encoder GEMM for batches several clips deep. PERSISTENT: one workgroup per CU walks its share of the
		// 256x256 output tiles. EIGHT waves as 2 (M) x 4 (N), a wave owns 128 x 64 outputs = 4 x 2 MFMA 32x32x16 tiles (24
		// fragment reads per 32 MFMAs; the 16-wave 64x64 layout of gemmTiled reads 16 per 16), both operands global -> LDS
		// directly in full 128-byte lines, XOR-swizzled source, two 64 KiB K-tile buffers + 32 KiB of epilogue staging = all
		// 160 KiB of a CU. What differs from gemmTiled is the SCHEDULE (cdna_hip_programming.md section 5, T3+T4):
		//   * a K tile is four phases, one 64x32 quadrant of the wave's outputs each (8 MFMAs = 256 matrix-pipe cycles):
		//       phase    fragments read from LDS        MFMAs          staged global -> LDS (1 KiB per instruction and wave)
		//       1        a0 (8 reads), b0 (4 reads)     a0 x b0        --
		//       2        b1 (4)                         a0 x b1        A rows   0..127 of K tile t+1 (2)
		//       3        a1 (8)                         a1 x b1        A rows 128..255 of K tile t+1 (2)
		//       4        --                             a1 x b0        W tile (256 rows) of K tile t+2 (4)
		//     every phase is  { ds_reads, LDS-DMA issue } s_barrier { MFMAs } s_barrier;
		//   * the two wave rows run ONE barrier apart (the waves of row 1 execute an extra s_barrier before the loop, those of
		//     row 0 after it): on every SIMD one wave is in its MFMA segment while the other reads fragments and issues DMA,
		//     so the matrix pipe never waits for a barrier, an LDS round trip or a DMA issue slot -- as long as a read/issue
		//     segment fits under 256 cycles, which is why the eight DMA instructions of a K tile are spread over three phases;
		//   * vmcnt never drops to 0 inside the loop: W is requested a whole K tile ahead (phase 4 of tile t for t+2), A as soon
		//     as its buffer half is dead (phases 2 and 3 of tile t for t+1); the waits sit at the end of phase 4's issue segment
		//     (vmcnt(6): W and the first A half of t+1) and of its MFMA segment (vmcnt(4): the second A half), so 24 .. 64 KiB per
		//     CU are in flight at any time and a DMA has 3 (A rows 128..), 4 (A rows 0..) or 8 (W) barrier intervals to land.
		//     Waiting per half tile three intervals after its issue (the first version) kept 16 .. 32 KiB in flight and was
		//     latency-bound at 54 GB/s per CU (profiles/r03_gemm8_ablation.txt);
		//   * DMA addresses are SGPR base (advanced per K tile on the scalar unit) + a per-lane 32-bit byte offset that never
		//     changes: no vector ALU work per instruction;
		//   * the NEXT tile's first operands are requested before this tile's epilogue starts, and the epilogue goes through its
		//     own 4 KiB per wave, so a tile's stores drain under the next tile's K loop and its first-tile latency under the epilogue.
		// Hazards (interval = barrier to barrier, K tile t occupies intervals 0..7 of wave row 0 and 1..8 of row 1):
		//   RAW  operands of tile t+1: W issued in -2 / -1 (row 0 / row 1), A rows 0.. in 2 / 3, A rows 128.. in 4 / 5. Row 0 reads
		//        W and A rows 0.. from interval 8, row 1 reads W and A rows 128.. from 9. Waits: vmcnt(6) at the end of 6 / 7
		//        (W, A rows 0..), vmcnt(4) at the end of 7 / 8 (A rows 128..); each is followed by a barrier both rows pass
		//        before the first read.
		//   WAR  buffer (t+1)&1 was last read by tile t-1: its W in interval -5 (row 1, phase 2), A rows 0.. in -4 (row 0, phase
		//        3), A rows 128.. in -3 (row 1, phase 3); those reads are retired by the MFMAs of the following interval, and the
		//        first DMA into each region is issued in -2, 2 and 4: at least two barriers later.
		struct Cfg8
		{
			static constexpr int BM = 256, BN = 256, BK = 64, NT = 512, TI = 4, TJ = 2;
			static constexpr int A_HALFS = BM * BK, STAGE = ( BM + BN ) * BK;	   // halfs per K-tile buffer: A tile, then W tile
			static constexpr int EPI_OFFSET = 2 * STAGE * 2;					   // bytes: the epilogue staging starts behind the two buffers
			static constexpr int EPI_PER_WAVE = 4096;
			static constexpr int LDS_BYTES = EPI_OFFSET + 8 * EPI_PER_WAVE;		   // 160 KiB
		};

		// Two 16-byte-per-lane global -> LDS instructions (2 x 1 KiB, LDS destinations dst and dst + 1024; global addresses
		// base + off0 / base + off1 with a wave-uniform 64-bit base). M0 (the LDS destination) is saved and restored inside the
		// statement: the compiler does not preserve it around inline assembly (cdna_hip_programming.md section 5.7).
		__device__ __forceinline__ void ldsDmaPair( const void* base, unsigned off0, unsigned off1, unsigned dst )
		{
			unsigned keep;
			// s_nop 1 / s_nop 0: wait states between the scalar writes (M0; a base computed just before the statement) and the
			// memory instruction that reads them -- nothing inside an asm string is padded by the compiler
			asm volatile( "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
						  "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\ts_mov_b32 m0, %0"
						  : "=&s"( keep )
						  : "s"( base ), "v"( off0 ), "v"( off1 ), "s"( dst ), "s"( dst + 1024u )
						  : "memory" );
		}
#define WH_BAR() asm volatile( "s_barrier" ::: "memory" )

		// One 32-row x 64-column block of a wave's outputs (MFMA tiles c0 = columns 0..31, c1 = 32..63 of the block) through 4 KiB of
		// LDS, leaving as 16-byte stores along the rows of the destination: the arithmetic of tileEpilogue / tileEpilogueWide per
		// element, FP16 outputs in one pass ([32][64] halfs), FP32 outputs in two ([16][64] floats each), 16-byte chunk index XORed
		// with the row so that the column-wise writes and the row-wise reads are both conflict free. m0 / n0 = first row / column.
		// Preconditions as for tileEpilogueWide (a.wideEpi). Residual / positional rows are requested before the LDS round trip.
		template<int EPI>
		__device__ __forceinline__ void epilogueBlock32x64( const GemmArgs& a, const f32x16& c0, const f32x16& c1, int m0, int n0, int lane, unsigned char* ldsWave )
		{
			const int hi = lane >> 5, c = lane & 31;
			float bias[ 2 ];
	#pragma unroll
			for( int j = 0; j < 2; j++ )
			{
				const int n = n0 + j * 32 + c;
				bias[ j ] = ( a.bias && n < a.N ) ? a.bias[ n ] : 0.0f;
			}
			if constexpr( EPI == EPI_F16_GELU || EPI == EPI_QKV_ENC || EPI == EPI_CROSS_KV )
			{
				f16* const L = (f16*)ldsWave;
				const int d = a.H * HEAD_DIM;
				int sel = 0, head = 0, layer = 0;
				if constexpr( EPI == EPI_QKV_ENC )
				{
					sel = n0 / d;
					head = ( n0 - sel * d ) >> 6;
				}
				if constexpr( EPI == EPI_CROSS_KV )
				{
					layer = n0 / ( 2 * d );
					const int c2 = n0 - layer * 2 * d;
					sel = c2 >= d ? 1 : 0;
					head = ( sel ? c2 - d : c2 ) >> 6;
				}
	#pragma unroll
				for( int j = 0; j < 2; j++ )
	#pragma unroll
					for( int r = 0; r < 16; r++ )
					{
						const int row = ( r & 3 ) + 8 * ( r >> 2 ) + 4 * hi;
						const int col = j * 32 + c;
						const float v = j == 0 ? c0[ r ] : c1[ r ];
						f16 hv;
						if constexpr( EPI == EPI_F16_GELU )
							hv = gelu16( v + bias[ j ] );
						else if constexpr( EPI == EPI_QKV_ENC )
							hv = (f16)( v + bias[ j ] );
						else
							hv = sel ? (f16)( v + bias[ j ] ) : (f16)( v * a.scale );
						L[ row * 64 + ( ( ( col >> 3 ) ^ ( row & 7 ) ) << 3 ) + ( col & 7 ) ] = hv;
					}
				__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
				const int chunk = lane & 7;
	#pragma unroll
				for( int it = 0; it < 4; it++ )
				{
					const int row = it * 8 + ( lane >> 3 );
					const int m = m0 + row;
					const f16x8 v = *(const f16x8*)( L + row * 64 + ( ( chunk ^ ( row & 7 ) ) << 3 ) );
					const int n = n0 + chunk * 8;
					if( m >= a.M || n >= a.N ) continue;
					if constexpr( EPI == EPI_F16_GELU )
						*(f16x8*)( a.out16 + rowOffset( m, a.Mb, a.ldc, a.cBatchStride ) + n ) = v;
					else
					{
						const int b = m / a.T;
						const int t = m - b * a.T;
						if constexpr( EPI == EPI_QKV_ENC )
						{
							f16* const dst = sel == 0 ? a.q : a.k;
							*(f16x8*)( dst + ( ( (long long)b * a.H + head ) * a.T + t ) * HEAD_DIM + chunk * 8 ) = v;
						}
						else
						{
							f16* const dst = sel ? a.v : a.k;
							*(f16x8*)( dst + ( ( ( (long long)layer * a.B + b ) * a.H + head ) * a.T + t ) * HEAD_DIM + chunk * 8 ) = v;
						}
					}
				}
				__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
			}
			else
			{
				static_assert( EPI == EPI_F32 || EPI == EPI_CONV2, "FP32 block epilogue" );
				float* const L = (float*)ldsWave;
				const int chunk = lane & 15;
				// everything the block READS from memory first: 2 halves x 4 rows x 16 bytes per lane
				f32x4 ex[ 2 ][ 4 ];
				long long off[ 2 ][ 4 ];
	#pragma unroll
				for( int hh = 0; hh < 2; hh++ )
	#pragma unroll
					for( int u = 0; u < 4; u++ )
					{
						int m = m0 + hh * 16 + u * 4 + ( lane >> 4 );
						m = m < a.M ? m : a.M - 1;
						int n = n0 + chunk * 4;
						n = n < a.N ? n : a.N - 4;
						if constexpr( EPI == EPI_F32 )
						{
							off[ hh ][ u ] = rowOffset( m, a.Mb, a.ldc, a.cBatchStride ) + n;
							ex[ hh ][ u ] = a.res ? *(const f32x4*)( a.res + off[ hh ][ u ] ) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
						}
						else
						{
							const int b = m / a.Mb;
							off[ hh ][ u ] = (long long)m * a.ldc + n;
							ex[ hh ][ u ] = *(const f32x4*)( a.pe + (long long)( m - b * a.Mb ) * a.N + n );
						}
					}
	#pragma unroll
				for( int hh = 0; hh < 2; hh++ )
				{
	#pragma unroll
					for( int j = 0; j < 2; j++ )
	#pragma unroll
						for( int q = 0; q < 8; q++ )
						{
							const int r = hh * 8 + q;
							const int row = ( q & 3 ) + 8 * ( q >> 2 ) + 4 * hi;	  // within the 16-row half
							const int col = j * 32 + c;
							float v = ( j == 0 ? c0[ r ] : c1[ r ] ) + bias[ j ];
							if constexpr( EPI == EPI_CONV2 ) v = (float)gelu16( v );
							L[ row * 64 + ( ( ( col >> 2 ) ^ row ) << 2 ) + ( col & 3 ) ] = v;
						}
					__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
	#pragma unroll
					for( int u = 0; u < 4; u++ )
					{
						const int row = u * 4 + ( lane >> 4 );
						const int m = m0 + hh * 16 + row;
						const int n = n0 + chunk * 4;
						const f32x4 v = *(const f32x4*)( L + row * 64 + ( ( chunk ^ row ) << 2 ) );
						if( m >= a.M || n >= a.N ) continue;
						f32x4 o;
	#pragma unroll
						for( int e = 0; e < 4; e++ ) o[ e ] = EPI == EPI_F32 ? v[ e ] + ex[ hh ][ u ][ e ] : ex[ hh ][ u ][ e ] + v[ e ];
						*(f32x4*)( a.out32 + off[ hh ][ u ] ) = o;
					}
					__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
				}
			}
		}

		// The V columns of the encoder's Q/K/V product, straight from the accumulators: fragment-major V (vFragIndex) keeps the
		// keys k..k+3 and k+8..k+11 of one dimension in one 16-byte chunk, and a lane of the 32x32 accumulator tile holds
		// exactly rows 4 hi + 8 g + {0..3} of one column -- so each group g of 4 registers is one 8-byte half of a chunk, and
		// groups g, g+1 are one whole chunk when the first one's key is a multiple of 8 inside its 16-key block. The chunks of
		// a store instruction are consecutive in memory (lane = dimension, hi = chunk + 32): 1 KiB per wave and instruction.
		// Requires T % 4 == 0 (a group of 4 rows never straddles two sequences); m0 / n0 = first row / column of the
		// 32 x 64 block, n0 a multiple of 64 inside the V third of the columns.
		__device__ __forceinline__ void epilogueBlockV32x64( const GemmArgs& a, const f32x16& c0, const f32x16& c1, int m0, int n0, int lane )
		{
			const int hi = lane >> 5, dd = lane & 31;
			const int d = a.H * HEAD_DIM;
			const int head = ( n0 - 2 * d ) >> 6;
			const int b0 = m0 / a.T;	   // wave-uniform
			const long long perSeq = (long long)a.H * HEAD_DIM * a.Tpad;
			f16* const vHead = a.v + (long long)head * HEAD_DIM * a.Tpad;
			int bOf[ 4 ], tOf[ 4 ];
			bool ok[ 4 ];
	#pragma unroll
			for( int g = 0; g < 4; g++ )
			{
				const int m = m0 + 4 * hi + 8 * g;
				int t = m - b0 * a.T, b = b0;
				if( t >= a.T )	  // the block runs into the next sequence (or, for T < 32, further)
				{
					b = m / a.T;
					t = m - b * a.T;
				}
				bOf[ g ] = b;
				tOf[ g ] = t;
				ok[ g ] = m < a.M;
			}
	#pragma unroll
			for( int j = 0; j < 2; j++ )
			{
				const int n = n0 + j * 32 + dd;
				const float bias = ( a.bias && n < a.N ) ? a.bias[ n ] : 0.0f;
				f16x4 pk[ 4 ];
	#pragma unroll
				for( int g = 0; g < 4; g++ )
	#pragma unroll
					for( int e = 0; e < 4; e++ ) pk[ g ][ e ] = (f16)( ( j == 0 ? c0[ 4 * g + e ] : c1[ 4 * g + e ] ) + bias );
				if( n >= a.N ) continue;
				auto dst = [ & ]( int g ) -> f16*
				{
					const int t = tOf[ g ];
					return vHead + bOf[ g ] * perSeq + ( ( (long long)( t >> 4 ) * 2 + j ) * 64 + ( ( t >> 2 ) & 1 ) * 32 + dd ) * 8 + ( ( t >> 3 ) & 1 ) * 4;
				};
				// groups g and g + 1 are one 16-byte chunk when g's keys are the first half of their 16-key block and g + 1 belongs
				// to the same sequence (a block that runs into the next sequence restarts the key count: checked per pair)
				auto whole = [ & ]( int g ) { return ok[ g + 1 ] && bOf[ g + 1 ] == bOf[ g ] && ( ( tOf[ g ] >> 3 ) & 1 ) == 0; };
				auto store16 = [ & ]( int g )
				{
					f16x8 w;
	#pragma unroll
					for( int e = 0; e < 4; e++ )
					{
						w[ e ] = pk[ g ][ e ];
						w[ 4 + e ] = pk[ g + 1 ][ e ];
					}
					*(f16x8*)dst( g ) = w;
				};
				auto store8 = [ & ]( int g )
				{
					if( ok[ g ] ) *(f16x4*)dst( g ) = pk[ g ];
				};
				if( whole( 0 ) )
				{
					store16( 0 );
					if( whole( 2 ) )
						store16( 2 );
					else
					{
						store8( 2 );
						store8( 3 );
					}
				}
				else
				{
					store8( 0 );
					if( whole( 1 ) )
					{
						store16( 1 );
						store8( 3 );
					}
					else
					{
						store8( 1 );
						if( whole( 2 ) )
							store16( 2 );
						else
						{
							store8( 2 );
							store8( 3 );
						}
					}
				}
			}
		}

		// the interior-tile epilogue of both persistent kernels (defined with gemmTiled4 below)
		template<int EPI, bool HASRES, int FIRST = 0, int LAST = 16, int TJ = 4, bool AGPR = true, bool L16 = false, typename ACC>
		__device__ __forceinline__ void epilogueFast4( const GemmArgs& a, ACC& acc, int mW, int nW, int lane, unsigned char* stage );

		// MF16 (round 6, the default: option gemm_mf16): the K loop on v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16 -- a quadrant is 4 x 2 tiles of
		// 16 x 16 over two k-halves of 32, the W fragment the srcB operand of four consecutive instructions; the same LDS image, the same 24 fragment reads per K
		// tile, the same 128 accumulator registers. The chip sustains more of this shape under its power limit (tools/mfma_order_probe.hip; the vendor library's
		// kernel uses it), and the SUMS ARE THE SAME BITS: the matrix cores add a k-block of 8 (one lane's 16 bytes) at a time in both shapes, and both kernels hand
		// them the k-blocks of a row in the same order (max |diff| = 0 against the 32x32x16 instance on every probed shape; the model-level identity test covers
		// it). Interior tiles leave through epilogueFast4's L16 form straight from the 16 x 16 tiles; edge tiles and the V third of the encoder's Q/K/V product are
		// first brought into the 32 x 32 register layout through the wave's staging area (convert16) and take the epilogues written for it.
		// Measured (profiles/r06_evidence/gemm_vendor_gap.txt): probe +3.4 .. 4.5 % on the encoder's shapes, the class in the model 0.379 -> 0.41 of the MFMA peak.
		template<int EPI, bool WIDE, bool MF16 = false>
		__global__ void __launch_bounds__( 512, 2 ) gemmTiled8( const GemmArgs a )
		{
			using C = Cfg8;
			constexpr int BM = C::BM, BN = C::BN, BK = C::BK;
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smem[];
			f16* const lds = (f16*)smem;
			typedef __attribute__( ( address_space( 3 ) ) ) void* LdsPtr;

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
			const int wr = wave >> 2, wc = wave & 3;

			// ---- this workgroup's tiles: XCD x (workgroup id % 8) owns a contiguous range of the band-walk order; its workgroups
			// take consecutive tiles of that range round by round, so the ~32 tiles an XCD has in flight are neighbours in the walk
			const int tilesM = ( a.M + BM - 1 ) / BM, tilesN = ( a.N + BN - 1 ) / BN;
			const int nTiles = tilesM * tilesN;
			const int gm = a.groupM;
			int linFirst, linEnd, linStep;
			{
				const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
				const int q = nTiles >> 3, r = nTiles & 7;
				const int start = xcd < r ? xcd * ( q + 1 ) : r * ( q + 1 ) + ( xcd - r ) * q;
				linEnd = start + ( xcd < r ? q + 1 : q );
				linFirst = start + idx;
				linStep = ( gridDim.x + 7 - xcd ) >> 3;	   // workgroups of this XCD
			}
			auto tileCoords = [ & ]( int lin, int& tm, int& tn )
			{
				if( gm > 1 )
				{
					const int perBand = gm * tilesN;
					const int band = lin / perBand;
					const int first = band * gm;
					const int rows = min( tilesM - first, gm );
					const int r = lin - band * perBand;
					tm = first + r % rows;
					tn = r / rows;
				}
				else
				{
					tm = lin / tilesN;
					tn = lin - tm * tilesN;
				}
			};

			// ---- LDS-DMA sources: a half tile is 128 rows x 128 bytes = 16 pieces of 8 rows, a wave owns pieces 2 wave, 2 wave + 1.
			// Lane l of a piece lands at row l / 8, physical 16-byte chunk l % 8, which must hold logical chunk (l % 8) ^ ((row >> 1) & 7).
			// offA / offW = byte offset of that chunk from a.A / a.W at k = 0 (the launcher guarantees they fit 32 bits).
			const int rIn = lane >> 3, cPhys = lane & 7;
			unsigned offA[ 2 ][ 2 ], offW[ 2 ][ 2 ];
			auto tileOffsets = [ & ]( int tm, int tn )
			{
	#pragma unroll
				for( int h = 0; h < 2; h++ )
	#pragma unroll
					for( int i = 0; i < 2; i++ )
					{
						const int row = h * 128 + ( wave * 2 + i ) * 8 + rIn;
						const int c = cPhys ^ ( ( row >> 1 ) & 7 );
						int m = tm * BM + row;
						m = m < a.M ? m : a.M - 1;
						offA[ h ][ i ] = (unsigned)( ( rowOffset( m, a.Mb, a.lda, a.aBatchStride ) + c * 8 ) * 2 );
						int n = tn * BN + row;
						n = n < a.N ? n : a.N - 1;
						offW[ h ][ i ] = (unsigned)( ( (long long)n * a.K + c * 8 ) * 2 );
					}
			};
			const unsigned ldsBase = __builtin_amdgcn_readfirstlane( (unsigned)(size_t)(LdsPtr)lds );
			// byte address of this wave's first piece of a half tile inside buffer 0: + buf * STAGE * 2, + (W ? A_HALFS * 2 : 0), + h * 16384
			const unsigned pieceBase = ldsBase + (unsigned)wave * 2048u;
			// part: 0 = W rows 0.., 1 = W rows 128.., 2 = A rows 0.., 3 = A rows 128..
			auto stage = [ & ]( int kt, auto part )
			{
				constexpr int P = decltype( part )::value;
				constexpr bool isW = P < 2;
				constexpr int h = P & 1;
				const unsigned dst = pieceBase + (unsigned)( kt & 1 ) * ( C::STAGE * 2 ) + ( isW ? C::A_HALFS * 2 : 0 ) + h * 16384;
				const f16* const base = ( isW ? a.W : a.A ) + kt * BK;
				if constexpr( isW )
					ldsDmaPair( base, offW[ h ][ 0 ], offW[ h ][ 1 ], dst );
				else
					ldsDmaPair( base, offA[ h ][ 0 ], offA[ h ][ 1 ], dst );
			};
			using PW0 = std::integral_constant<int, 0>;
			using PW1 = std::integral_constant<int, 1>;
			using PA0 = std::integral_constant<int, 2>;
			using PA1 = std::integral_constant<int, 3>;
			const int nk = a.K / BK;
			// first operands of a tile: K tile 0 into buffer 0 and the W tile of K tile 1 into buffer 1 (12 instructions per wave)
			auto stageFirst = [ & ]()
			{
				stage( 0, PW0{} );
				stage( 0, PW1{} );
				stage( 0, PA0{} );
				stage( 0, PA1{} );
				if( nk > 1 )
				{
					stage( 1, PW0{} );
					stage( 1, PW1{} );
				}
			};

			// ---- fragment reads: lane l reads row l & 31 of a 32-row tile, logical chunk 2 ks + (l >> 5), stored at chunk ^ ((row >> 1) & 7);
			// the tile origins are multiples of 32 rows, so the XOR term depends on the lane only
			const int x0 = ( lane >> 5 ) ^ ( ( lane >> 1 ) & 7 );
			int laneK[ 4 ];
	#pragma unroll
			for( int ks = 0; ks < 4; ks++ ) laneK[ ks ] = ( lane & 31 ) * BK + ( ( x0 ^ ( ks << 1 ) ) << 3 );
			// MF16: lane l reads row l & 15 of a 16-row tile, logical chunk 4 h + (l >> 4) of k-half h
			int laneK16[ 2 ];
#pragma unroll
			for( int h = 0; h < 2; h++ ) laneK16[ h ] = ( lane & 15 ) * BK + ( ( ( ( h << 2 ) + ( lane >> 4 ) ) ^ ( ( lane >> 1 ) & 7 ) ) << 3 );
			const int aRow0 = wr * 128, wRow0 = wc * 64;

			int tm, tn;
			int lin = linFirst;
			if( lin >= linEnd ) return;
			tileCoords( lin, tm, tn );
			tileOffsets( tm, tn );
			stageFirst();

			for( ;; )
			{
				f32x16 acc[ 4 ][ 2 ];
				f32x4 acc16[ MF16 ? 8 : 1 ][ MF16 ? 4 : 1 ];
				if constexpr( MF16 )
				{
	#pragma unroll
					for( int i = 0; i < 8; i++ )
	#pragma unroll
						for( int j = 0; j < 4; j++ )
	#pragma unroll
							for( int r = 0; r < 4; r++ ) acc16[ i ][ j ][ r ] = 0.0f;
				}
				else
				{
	#pragma unroll
					for( int i = 0; i < 4; i++ )
	#pragma unroll
						for( int j = 0; j < 2; j++ )
	#pragma unroll
							for( int r = 0; r < 16; r++ ) acc[ i ][ j ][ r ] = 0.0f;
				}
				// MF16: fa[ i' >> 1 ][ 2 ( i' & 1 ) + h ] = rows 16 i' of the half, k-half h; fb[ 2 j' + h ] = columns 16 j' of the 32, k-half h
				f16x8 fa[ 2 ][ 4 ], fb0[ 4 ], fb1[ 4 ];
				auto readA = [ & ]( const f16* bufA, int half )
				{
					if constexpr( MF16 )
					{
	#pragma unroll
						for( int ip = 0; ip < 4; ip++ )
	#pragma unroll
							for( int h = 0; h < 2; h++ )
								fa[ ip >> 1 ][ ( ( ip & 1 ) << 1 ) + h ] = *(const f16x8*)( bufA + ( aRow0 + half * 64 + ip * 16 ) * BK + laneK16[ h ] );
					}
					else
					{
	#pragma unroll
						for( int i = 0; i < 2; i++ )
	#pragma unroll
							for( int ks = 0; ks < 4; ks++ )
								fa[ i ][ ks ] = *(const f16x8*)( bufA + ( aRow0 + ( half * 2 + i ) * 32 ) * BK + laneK[ ks ] );
					}
				};
				auto readB = [ & ]( const f16* bufW, int j, f16x8( &fb )[ 4 ] )
				{
					if constexpr( MF16 )
					{
	#pragma unroll
						for( int jp = 0; jp < 2; jp++ )
	#pragma unroll
							for( int h = 0; h < 2; h++ ) fb[ ( jp << 1 ) + h ] = *(const f16x8*)( bufW + ( wRow0 + j * 32 + jp * 16 ) * BK + laneK16[ h ] );
					}
					else
					{
	#pragma unroll
						for( int ks = 0; ks < 4; ks++ ) fb[ ks ] = *(const f16x8*)( bufW + ( wRow0 + j * 32 ) * BK + laneK[ ks ] );
					}
				};
				auto quadrant = [ & ]( auto i0c, auto jc, const f16x8( &fb )[ 4 ] )
				{
					constexpr int i0 = decltype( i0c )::value, j = decltype( jc )::value;
					__builtin_amdgcn_s_setprio( 1 );
					if constexpr( MF16 )
					{
	#pragma unroll
						for( int h = 0; h < 2; h++ )
	#pragma unroll
							for( int jp = 0; jp < 2; jp++ )
	#pragma unroll
								for( int ip = 0; ip < 4; ip++ )
									acc16[ 2 * i0 + ip ][ 2 * j + jp ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( fa[ ip >> 1 ][ ( ( ip & 1 ) << 1 ) + h ], fb[ ( jp << 1 ) + h ],
										acc16[ 2 * i0 + ip ][ 2 * j + jp ], 0, 0, 0 );
					}
					else
					{
	#pragma unroll
						for( int ks = 0; ks < 4; ks++ )
	#pragma unroll
							for( int i = 0; i < 2; i++ )
								acc[ i0 + i ][ j ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( fa[ i ][ ks ], fb[ ks ], acc[ i0 + i ][ j ], 0, 0, 0 );
					}
					__builtin_amdgcn_s_setprio( 0 );
				};
				using I0 = std::integral_constant<int, 0>;
				using I1 = std::integral_constant<int, 1>;
				using I2 = std::integral_constant<int, 2>;

				// the tile's first operands were requested before the previous tile's epilogue (or above): K tile 0 must have landed
				if( nk > 1 )
					asm volatile( "s_waitcnt vmcnt(4)" ::: "memory" );
				else
					asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
				WH_BAR();
				if( wr == 1 ) WH_BAR();	   // wave row 1 runs one barrier behind row 0

				for( int kt = 0; kt < nk; kt++ )
				{
					const f16* const bufA = lds + ( kt & 1 ) * C::STAGE;
					const f16* const bufW = bufA + C::A_HALFS;
					const bool next = kt + 1 < nk, next2 = kt + 2 < nk;
					// phase 1: 12 fragment reads
					readB( bufW, 0, fb0 );
					readA( bufA, 0 );
					WH_BAR();
					quadrant( I0{}, I0{}, fb0 );
					WH_BAR();
					// phase 2: 4 reads, A rows 0..127 of K tile t+1
					readB( bufW, 1, fb1 );
					if( next ) stage( kt + 1, PA0{} );
					WH_BAR();
					quadrant( I0{}, I1{}, fb1 );
					WH_BAR();
					// phase 3: 8 reads, A rows 128..255 of K tile t+1
					readA( bufA, 1 );
					if( next ) stage( kt + 1, PA1{} );
					WH_BAR();
					quadrant( I2{}, I1{}, fb1 );
					WH_BAR();
					// phase 4: no reads, the W tile of K tile t+2; W and A rows 0.. of tile t+1 must have landed after the issue
					// segment, A rows 128.. after the MFMA segment
					if( next2 )
					{
						stage( kt + 2, PW0{} );
						stage( kt + 2, PW1{} );
						asm volatile( "s_waitcnt vmcnt(6)" ::: "memory" );
					}
					else if( next )
						asm volatile( "s_waitcnt vmcnt(2)" ::: "memory" );
					WH_BAR();
					quadrant( I2{}, I0{}, fb0 );
					if( next2 )
						asm volatile( "s_waitcnt vmcnt(4)" ::: "memory" );
					else
						asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
					WH_BAR();
				}
				if( wr == 0 ) WH_BAR();
				// every wave has passed the same number of barriers and retired all its fragment reads: both operand buffers are dead

				const int tmDone = tm, tnDone = tn;
				lin += linStep;
				const bool more = lin < linEnd;
				if( more )
				{
					tileCoords( lin, tm, tn );
					tileOffsets( tm, tn );
					stageFirst();	  // lands under the epilogue below
				}

				auto convert16 = [ & ]()
				{
					if constexpr( MF16 )
					{
						// 16 x 16 tiles -> the 32 x 32 register layout the general epilogues are written for (edge tiles, V tiles), one 32 x 32 block at a time through
						// the wave's 4 KiB (a wave's LDS operations execute in order: no wait between the writes, the reads and the next block's writes)
						float* const st = (float*)( smem + C::EPI_OFFSET + wave * C::EPI_PER_WAVE );
						const int q = lane >> 4, c16 = lane & 15, hi = lane >> 5, cl = lane & 31;
		#pragma unroll
						for( int i = 0; i < 4; i++ )
		#pragma unroll
							for( int j = 0; j < 2; j++ )
							{
		#pragma unroll
								for( int ti = 0; ti < 2; ti++ )
		#pragma unroll
									for( int tj = 0; tj < 2; tj++ )
		#pragma unroll
										for( int r = 0; r < 4; r++ ) st[ ( 16 * ti + 4 * q + r ) * 32 + 16 * tj + c16 ] = acc16[ 2 * i + ti ][ 2 * j + tj ][ r ];
								__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
								__builtin_amdgcn_wave_barrier();
								__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
		#pragma unroll
								for( int r = 0; r < 16; r++ ) acc[ i ][ j ][ r ] = st[ ( ( r & 3 ) + 8 * ( r >> 2 ) + 4 * hi ) * 32 + cl ];
								__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
								__builtin_amdgcn_wave_barrier();
								__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
							}

					}
				};

				bool direct = !WIDE;
				bool fastDone = false;
				if constexpr( WIDE && ( EPI == EPI_F32 || EPI == EPI_F16_GELU || EPI == EPI_QKV_ENC || EPI == EPI_CROSS_KV ) )
				{
					// interior tiles: the lean epilogue written for gemmTiled4 (no bounds checks, no divisions per row, residual rows requested a unit
					// ahead of the stores); a.wideEpi == 2 = the launcher has checked what it relies on
					const int mW = tmDone * BM + wr * 128, nW = tnDone * BN + wc * 64;
					const bool isV = EPI == EPI_QKV_ENC && nW >= 2 * a.H * HEAD_DIM;
					if( a.wideEpi == 2 && !isV && ( tmDone + 1 ) * BM <= a.M && ( tnDone + 1 ) * BN <= a.N )
					{
						unsigned char* const stage = smem + C::EPI_OFFSET + wave * C::EPI_PER_WAVE;
						if constexpr( MF16 )
						{
							if constexpr( EPI == EPI_F32 )
							{
								if( a.res )
									epilogueFast4<EPI, true, 0, 16, 2, false, true>( a, acc16, mW, nW, lane, stage );
								else
									epilogueFast4<EPI, false, 0, 16, 2, false, true>( a, acc16, mW, nW, lane, stage );
							}
							else
								epilogueFast4<EPI, false, 0, 16, 2, false, true>( a, acc16, mW, nW, lane, stage );
						}
						else if constexpr( EPI == EPI_F32 )
						{
							if( a.res )
								epilogueFast4<EPI, true, 0, 16, 2, false>( a, acc, mW, nW, lane, stage );
							else
								epilogueFast4<EPI, false, 0, 16, 2, false>( a, acc, mW, nW, lane, stage );
						}
						else
							epilogueFast4<EPI, false, 0, 16, 2, false>( a, acc, mW, nW, lane, stage );
						fastDone = true;
					}
				}
				if constexpr( MF16 )
				{
					if( !fastDone ) convert16();
				}
				if( fastDone )
				{
				}
				else
				if constexpr( WIDE && EPI == EPI_QKV_ENC )
				{
					// fragment-major V: straight from the registers (groups of 4 consecutive keys; T % 4 != 0 keeps the element-wise path)
					if( ( tnDone * BN + wc * 64 ) >= 2 * a.H * HEAD_DIM )
					{
						direct = ( a.T & 3 ) != 0;
						if( !direct )
						{
	#pragma unroll
							for( int i = 0; i < 4; i++ )
								epilogueBlockV32x64( a, acc[ i ][ 0 ], acc[ i ][ 1 ], tmDone * BM + wr * 128 + i * 32, tnDone * BN + wc * 64, lane );
						}
					}
				}
				if( fastDone )
				{
				}
				else if( direct )
					tileEpilogue<EPI, Cfg8>( a, acc, tmDone, tnDone, wr, wc, lane );
				else if( !( WIDE && EPI == EPI_QKV_ENC && ( tnDone * BN + wc * 64 ) >= 2 * a.H * HEAD_DIM ) )
				{
					if constexpr( WIDE && ( EPI == EPI_F32 || EPI == EPI_F16_GELU || EPI == EPI_CONV2 || EPI == EPI_QKV_ENC || EPI == EPI_CROSS_KV ) )
					{
#pragma unroll
						for( int i = 0; i < 4; i++ )
							epilogueBlock32x64<EPI>( a, acc[ i ][ 0 ], acc[ i ][ 1 ], tmDone * BM + wr * 128 + i * 32, tnDone * BN + wc * 64, lane,
								smem + C::EPI_OFFSET + wave * C::EPI_PER_WAVE );
					}
				}
				if( !more ) break;
			}
		}

		// ---------------------------------------------------------------------------------------------------------------
		// gemmTiled4 (round 4): the encoder product with ONE wave per SIMD -- the tile shape of the vendor library's kernel for these
		// shapes (profiles/r04_gemm_counters.txt). gemmTiled8's waves own 128 x 64 outputs, so every fragment read feeds half the MFMAs
		// it could (24 ds_read_b128 per 32 MFMAs). Here FOUR waves (2 x 2) own 128 x 128 each = 4 x 4 tiles of
		// v_mfma_f32_32x32x16_f16: 256 accumulator registers (the AGPR half of a 512-register wave), 8 fragment reads per 16 MFMAs,
		// half the LDS traffic per FLOP. With a single wave per SIMD nothing overlaps by itself, so the K loop is a software pipeline
		// written out by hand:
		//   * a K tile (64) is four substeps of 16 MFMAs in chunks of 4 (one A row tile x the four W tiles); the fragments of substep
		//     s + 1 are read (8 x ds_read_b128, second register set) behind the first 8 MFMAs of substep s, one read per MFMA
		//     (sched_group_barrier), and nothing crosses a chunk boundary (sched_barrier);
		//   * ONE s_barrier per K tile, before the LAST substep: by then a wave has read everything it needs from the current
		//     buffer (the last substep's fragments are in registers) and waited for its own LDS-DMA pieces of the next tile
		//     (vmcnt), so after the barrier the next K tile is complete in the other buffer and the current buffer is dead:
		//     the first fragments of the next K tile are read under the last substep's MFMAs and the DMA of the tile after
		//     next starts into the dead buffer, one pair of 1 KiB pieces behind each chunk of substeps 3 and 0. A piece has a whole
		//     K tile (~2k cycles) to land; the matrix pipe sees the barrier only as the skew between four waves that run the same stream;
		//   * the stream of K tiles is FLAT across output tiles (persistent workgroup, the band walk of gemmTiled8): the
		//     producer side (tile coordinates, per-lane source offsets, recomputed without a branch or a division per row) runs two
		//     K tiles ahead of the consumer and simply moves on to the next output tile; the epilogue of a tile runs between two K
		//     tiles with the next output tile's first two K tiles requested before its first store;
		//   * a tile's first substep multiplies into the constant 0 instead of clearing 256 registers;
		//   * ONE instance of every K tile position (first / middle / last) in a row and the epilogue outside the K loop: accumulators
		//     that meet at the end of alternative paths (a switch, a peeled variant) are 256 registers the allocator copies around.
		// LDS: two 64 KiB K-tile buffers (A rows, then W rows, 128-byte rows, 16-byte chunks XOR-swizzled exactly as gemmTiled8)
		// + 4 KiB of epilogue staging per wave = 144 KiB. Wave w stages rows 64 w .. 64 w + 63 of both operand tiles.
		//
		// MEASURED (MI355X, profiles/r04_gemm4_probe.txt): correct (bit-identical to gemmTiled8) and +7 .. 15 % on the plain FP32 probe
		// (168000 x 4096 x 1024: 930 against 850 TFLOP/s), but inside the model it is level with gemmTiled8 (GEMM class -2 % .. +0.3 %:
		// Q/K/V -5 %, GELU and cross-K/V +4 .. 5 %), so TUNE_GEMM_4WAVE is OFF. What it did settle, by ablation: without LDS-DMA and
		// without epilogue the K loop runs at 1500 TFLOP/s; the DMA costs 18 % of that whatever its placement (staggered over the waves,
		// spread over 2 or 3 substeps: the same) -- it is the CU's L2 -> LDS path, ~19 bytes per cycle under the MFMAs (26 alone), and a
		// 256 x 256 x 64 tile needs 64 KiB per 2048 matrix-pipe cycles = 32; the epilogue costs another 25 %: 2.5 us of instructions and
		// 4 .. 7 us in which the tile's 128 .. 256 KiB drain at the ~16 bytes per cycle a CU stores, with the next tile's DMA queued
		// behind them. Its lean epilogue, which does not depend on the wave shape, is what gemmTiled8 now uses (TUNE_GEMM_FAST_EPI).
		// An accumulator register of gemmTiled4 read where the epilogue uses it. Written as assembly so that the register allocator keeps
		// the 256 accumulators in the AGPR half of the file until then: left to itself it copies half of them into VGPRs at the end of
		// the K loop, spills the K loop's own state to scratch to make room, and every scratch reload then waits for ALL stores in
		// flight (vmcnt(0)). The MFMAs that wrote the accumulators are dozens of instructions behind the first read (the caller
		// computes the next tile's offsets in between and pads with s_nop): no hazard the compiler would have had to see.
		__device__ __forceinline__ f32x16 accReadTile( const f32x16& t )
		{
			f32x16 v;
	#pragma unroll
			for( int r = 0; r < 16; r++ )
			{
				float x;
				asm volatile( "v_accvgpr_read_b32 %0, %1" : "=v"( x ) : "a"( t[ r ] ) );
				v[ r ] = x;
			}
			return v;
		}

		// Epilogue of an INTERIOR 128 x 128 wave tile of gemmTiled4 (whole tile inside M x N; the launcher has checked what a.wideEpi == 2
		// promises below). One wave per SIMD: nothing hides a stall, so this path has no bounds checks, no divisions per row, no
		// branches, and an order of memory operations that never waits for a store:
		//   * the tile leaves in UNITS of 32 rows x 128 bytes (FP32: one MFMA tile; FP16: two side by side) through 4 KiB of LDS per
		//     wave: 16 / 32 column-wise writes per lane, then 4 x (ds_read_b128 -> 16-byte row store), 8 lanes per 128-byte row;
		//     LDS operations of a wave execute in order, so one buffer is enough and nothing but the data dependence is waited for;
		//   * software pipeline over the units, in program order: reads of unit k issued | residual rows of unit k + 1 requested |
		//     unit k + 1 converted and written to LDS (the GELU arithmetic sits here, under the LDS round trip of unit k) | unit k
		//     stored. A residual load is always older than the stores issued after it: waiting for it never waits for a store;
		//   * addresses are a scalar base per unit + one 32-bit offset per lane and row (16 registers for the 16 rows a lane stores,
		//     computed once per tile); a row past the end of its segment (sequence / conv batch) adds one constant: the wave's 128
		//     rows cross at most one boundary (segments are at least 128 rows long).
		// Same arithmetic per element as tileEpilogue / epilogueBlock32x64 (bit-identical outputs).
		// TJ = MFMA tiles per wave in N: 4 (gemmTiled4: 128 x 128 per wave) or 2 (gemmTiled8: 128 x 64); AGPR = the accumulators are read as assembly (gemmTiled4)
		// L16 (round 6): acc is f32x4[ 8 ][ 2 TJ ], the tiles of v_mfma_f32_16x16x32_f16 (lane l: column l & 15, rows 4 (l >> 4) .. + 3 of a 16 x 16 tile). Only the
		// column-wise writes into the staging area differ: a unit is the same 32 rows x 128 bytes, everything behind the LDS round trip is shared. The four row
		// groups of a tile (l >> 4) would meet in the same banks, so the 16-byte chunk index is XORed with a function of the row on both sides of the round trip.
		template<int EPI, bool HASRES, int FIRST, int LAST, int TJ, bool AGPR, bool L16, typename ACC>
		__device__ __forceinline__ void epilogueFast4( const GemmArgs& a, ACC& acc, int mW, int nW, int lane, unsigned char* stage )
		{
			static_assert( EPI == EPI_F32 || EPI == EPI_F16_GELU || EPI == EPI_QKV_ENC || EPI == EPI_CROSS_KV, "fast epilogue" );
			constexpr bool F32OUT = EPI == EPI_F32;
			constexpr bool HEADS = EPI == EPI_QKV_ENC || EPI == EPI_CROSS_KV;
			constexpr int UNITS = F32OUT ? 4 * TJ : 2 * TJ;
			constexpr int JP = TJ / 2;
			// (opaque copy: what follows is a few VALU instructions per tile; hoisted out of the tile loop it would live in scratch)
			asm volatile( "" : "+v"( lane ) );
			const int hi = lane >> 5, cl = lane & 31, rl = lane >> 3, ch = lane & 7;
			const int q16 = lane >> 4, c16 = lane & 15;
			float bias[ L16 ? 2 * TJ : TJ ];
			if constexpr( L16 )
			{
	#pragma unroll
				for( int j = 0; j < 2 * TJ; j++ ) bias[ j ] = a.bias ? a.bias[ nW + 16 * j + c16 ] : 0.0f;
			}
			else
			{
	#pragma unroll
				for( int j = 0; j < TJ; j++ ) bias[ j ] = a.bias ? a.bias[ nW + 32 * j + cl ] : 0.0f;
			}

			// ---- rows (wave-uniform): segment length, position of the tile's first row in its segment, byte offset of that row
			int seg, segPos;
			unsigned rowBytes, crossBytes;
			long long firstRowBytes;
			if constexpr( HEADS )
			{
				seg = a.T;
				const int b = mW / a.T;
				segPos = mW - b * a.T;
				rowBytes = 128u;
				crossBytes = (unsigned)( a.H - 1 ) * (unsigned)a.T * 128u;
				firstRowBytes = ( (long long)b * a.H * a.T + segPos ) * 128;
			}
			else
			{
				constexpr int ES = F32OUT ? 4 : 2;
				const int b = a.Mb > 0 ? mW / a.Mb : 0;
				seg = a.Mb > 0 ? a.Mb : 0x7fffffff;
				segPos = mW - b * ( a.Mb > 0 ? a.Mb : 0 );
				rowBytes = (unsigned)a.ldc * ES;
				crossBytes = (unsigned)( ( a.cBatchStride - (long long)a.Mb * a.ldc ) * ES );
				firstRowBytes = ( (long long)b * a.cBatchStride + (long long)segPos * a.ldc ) * ES;
			}
			seg = __builtin_amdgcn_readfirstlane( seg );
			segPos = __builtin_amdgcn_readfirstlane( segPos );
			rowBytes = __builtin_amdgcn_readfirstlane( rowBytes );
			crossBytes = __builtin_amdgcn_readfirstlane( crossBytes );
			unsigned voff[ 4 ][ 4 ];
	#pragma unroll
			for( int i = 0; i < 4; i++ )
	#pragma unroll
				for( int it = 0; it < 4; it++ )
				{
					const int r = 32 * i + 8 * it + rl;
					voff[ i ][ it ] = (unsigned)r * rowBytes + ( segPos + r >= seg ? crossBytes : 0u ) + (unsigned)ch * 16u;
				}

			// ---- columns (wave-uniform): what the wave's 128 columns are, base address of unit k
			int sel = 0;
			long long colBytes = 0;	   // byte offset of the wave tile's first column block
			if constexpr( EPI == EPI_F32 ) colBytes = (long long)nW * 4;
			if constexpr( EPI == EPI_F16_GELU ) colBytes = (long long)nW * 2;
			if constexpr( EPI == EPI_QKV_ENC )
			{
				const int d = a.H * HEAD_DIM;
				sel = __builtin_amdgcn_readfirstlane( nW / d );
				colBytes = (long long)( ( nW - sel * d ) >> 6 ) * a.T * 128;
			}
			if constexpr( EPI == EPI_CROSS_KV )
			{
				const int d = a.H * HEAD_DIM;
				const int layer = __builtin_amdgcn_readfirstlane( nW / ( 2 * d ) );
				const int c2 = nW - layer * 2 * d;
				sel = c2 >= d ? 1 : 0;
				colBytes = ( (long long)layer * a.B * a.H + ( ( sel ? c2 - d : c2 ) >> 6 ) ) * a.T * 128;
			}
			unsigned char* outBase;
			if constexpr( EPI == EPI_F32 ) outBase = (unsigned char*)a.out32;
			if constexpr( EPI == EPI_F16_GELU ) outBase = (unsigned char*)a.out16;
			if constexpr( EPI == EPI_QKV_ENC ) outBase = (unsigned char*)( sel == 0 ? a.q : a.k );
			if constexpr( EPI == EPI_CROSS_KV ) outBase = (unsigned char*)( sel ? a.v : a.k );
			outBase += firstRowBytes + colBytes;
			const unsigned char* resBase = HASRES ? (const unsigned char*)a.res + firstRowBytes + colBytes : nullptr;
			// bytes from the wave tile's first unit to unit k: FP32 unit k = MFMA tile (k / TJ, k % TJ); FP16 unit k = tiles (k / JP, 2 (k % JP)), (.., + 1)
			const long long headBytes = HEADS ? (long long)a.T * 128 : 128;

			auto writeUnit = [ & ]( auto kc )
			{
				constexpr int k = decltype( kc )::value;
				if constexpr( L16 )
				{
					// row R = 16 ti + 4 q + r of the unit; its chunk index is XORed with swz( R ) = ((R >> 2) & 1) << 2 (FP32: eight 4-column chunks per row) or
					// ((R >> 2) & 3) << 1 (FP16: eight 8-column chunks); (R >> 2) & 3 = q for every ti and r
					if constexpr( F32OUT )
					{
						constexpr int i = k / TJ, j = k % TJ;
						const int sw = ( q16 & 1 ) << 2;
	#pragma unroll
						for( int ti = 0; ti < 2; ti++ )
	#pragma unroll
							for( int tj = 0; tj < 2; tj++ )
	#pragma unroll
								for( int r = 0; r < 4; r++ )
								{
									const int row = 16 * ti + 4 * q16 + r;
									const int chunk = ( 4 * tj + ( c16 >> 2 ) ) ^ sw;
									*(float*)( stage + row * 128 + chunk * 16 + ( c16 & 3 ) * 4 ) = acc[ 2 * i + ti ][ 2 * j + tj ][ r ] + bias[ 2 * j + tj ];
								}
					}
					else
					{
						constexpr int i = k / JP, jp = k % JP;
						const int sw = q16 << 1;
	#pragma unroll
						for( int ti = 0; ti < 2; ti++ )
	#pragma unroll
							for( int tj = 0; tj < 4; tj++ )
	#pragma unroll
								for( int r = 0; r < 4; r++ )
								{
									const int row = 16 * ti + 4 * q16 + r;
									const int chunk = ( 2 * tj + ( c16 >> 3 ) ) ^ sw;
									const float v = acc[ 2 * i + ti ][ 4 * jp + tj ][ r ];
									const float b = bias[ 4 * jp + tj ];
									f16 hv;
									if constexpr( EPI == EPI_F16_GELU )
										hv = gelu16( v + b );
									else if constexpr( EPI == EPI_QKV_ENC )
										hv = (f16)( v + b );
									else
										hv = sel ? (f16)( v + b ) : (f16)( v * a.scale );
									*(f16*)( stage + row * 128 + chunk * 16 + ( c16 & 7 ) * 2 ) = hv;
								}
					}
				}
				else if constexpr( F32OUT )
				{
					constexpr int i = k / TJ, j = k % TJ;
	#pragma unroll
					for( int r = 0; r < 16; r++ )
					{
						const int row = ( r & 3 ) + 8 * ( r >> 2 ) + 4 * hi;
						float x;
						if constexpr( AGPR )
							asm volatile( "v_accvgpr_read_b32 %0, %1" : "=v"( x ) : "a"( acc[ i ][ j ][ r ] ) );
						else
							x = acc[ i ][ j ][ r ];
						*(float*)( stage + row * 128 + cl * 4 ) = x + bias[ j ];
					}
				}
				else
				{
					constexpr int i = k / JP, jp = k % JP;
	#pragma unroll
					for( int jj = 0; jj < 2; jj++ )
	#pragma unroll
						for( int r = 0; r < 16; r++ )
						{
							const int row = ( r & 3 ) + 8 * ( r >> 2 ) + 4 * hi;
							float v;
							if constexpr( AGPR )
								asm volatile( "v_accvgpr_read_b32 %0, %1" : "=v"( v ) : "a"( acc[ i ][ 2 * jp + jj ][ r ] ) );
							else
								v = acc[ i ][ 2 * jp + jj ][ r ];
							f16 hv;
							if constexpr( EPI == EPI_F16_GELU )
								hv = gelu16( v + bias[ 2 * jp + jj ] );
							else if constexpr( EPI == EPI_QKV_ENC )
								hv = (f16)( v + bias[ 2 * jp + jj ] );
							else
								hv = sel ? (f16)( v + bias[ 2 * jp + jj ] ) : (f16)( v * a.scale );
							*(f16*)( stage + row * 128 + ( jj * 32 + cl ) * 2 ) = hv;
						}
				}
			};
			auto ldsFence = [ & ]()
			{
				// compile-time only: the column-wise writes and the row-wise reads of the staging area use different types
				__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
			};
			auto unitBytes = [ & ]( int k ) -> long long { return F32OUT ? (long long)( k % TJ ) * 128 : (long long)( k % JP ) * headBytes; };
			auto loadRes = [ & ]( auto kc, f32x4( &ex )[ 4 ] )
			{
				constexpr int k = decltype( kc )::value;
				if constexpr( HASRES )
				{
					constexpr int i = F32OUT ? k / TJ : k / JP;
					const unsigned char* const b = resBase + unitBytes( k );
	#pragma unroll
					for( int it = 0; it < 4; it++ ) ex[ it ] = *(const f32x4*)( b + voff[ i ][ it ] );
				}
			};
			auto readUnit = [ & ]( f32x4( &dv )[ 4 ] )
			{
	#pragma unroll
				for( int it = 0; it < 4; it++ )
				{
					int chunk = ch;
					if constexpr( L16 )
					{
						const int row = it * 8 + rl;
						chunk = F32OUT ? ( ch ^ ( ( ( row >> 2 ) & 1 ) << 2 ) ) : ( ch ^ ( ( ( row >> 2 ) & 3 ) << 1 ) );
					}
					dv[ it ] = *(const f32x4*)( stage + ( it * 8 + rl ) * 128 + chunk * 16 );
				}
			};
			auto storeUnit = [ & ]( auto kc, const f32x4( &dv )[ 4 ], const f32x4( &ex )[ 4 ] )
			{
				constexpr int k = decltype( kc )::value;
				constexpr int i = F32OUT ? k / TJ : k / JP;
				unsigned char* const b = outBase + unitBytes( k );
	#pragma unroll
				for( int it = 0; it < 4; it++ )
				{
					f32x4 o = dv[ it ];
					if constexpr( HASRES )
					{
	#pragma unroll
						for( int e = 0; e < 4; e++ ) o[ e ] = dv[ it ][ e ] + ex[ it ][ e ];
					}
					*(f32x4*)( b + voff[ i ][ it ] ) = o;
				}
			};

			// units FIRST .. min( LAST, UNITS ) - 1 (gemmTiled4 keeps the rest of an FP16 tile in registers and stores it under the next tile's K loop)
			constexpr int U0 = FIRST, U1 = LAST < UNITS ? LAST : UNITS;
			if constexpr( U0 < U1 )
			{
				f32x4 ex[ 2 ][ 4 ], dv[ 4 ];
				loadRes( std::integral_constant<int, U0>{}, ex[ U0 & 1 ] );
				writeUnit( std::integral_constant<int, U0>{} );
				ldsFence();
				__builtin_amdgcn_sched_barrier( 0 );
				auto step = [ & ]( auto kc )
				{
					constexpr int k = decltype( kc )::value;
					if constexpr( k >= U0 && k < U1 )
					{
						readUnit( dv );
						ldsFence();
						__builtin_amdgcn_sched_barrier( 0 );
						if constexpr( k + 1 < U1 )
						{
							loadRes( std::integral_constant<int, k + 1>{}, ex[ ( k + 1 ) & 1 ] );
							writeUnit( std::integral_constant<int, k + 1>{} );
							ldsFence();
							__builtin_amdgcn_sched_barrier( 0 );
						}
						storeUnit( kc, dv, ex[ k & 1 ] );
						__builtin_amdgcn_sched_barrier( 0 );
					}
				};
				step( std::integral_constant<int, 0>{} );
				step( std::integral_constant<int, 1>{} );
				step( std::integral_constant<int, 2>{} );
				step( std::integral_constant<int, 3>{} );
				step( std::integral_constant<int, 4>{} );
				step( std::integral_constant<int, 5>{} );
				step( std::integral_constant<int, 6>{} );
				step( std::integral_constant<int, 7>{} );
				step( std::integral_constant<int, 8>{} );
				step( std::integral_constant<int, 9>{} );
				step( std::integral_constant<int, 10>{} );
				step( std::integral_constant<int, 11>{} );
				step( std::integral_constant<int, 12>{} );
				step( std::integral_constant<int, 13>{} );
				step( std::integral_constant<int, 14>{} );
				step( std::integral_constant<int, 15>{} );
			}
		}

		// The V third of the encoder's Q/K/V product, interior wave tile of gemmTiled4: fragment-major V (vFragIndex) straight from the
		// accumulators, no LDS. A lane of a 32x32 accumulator tile holds one dimension and, per register group g, the 4 consecutive keys
		// t .. t + 3 (t % 4 == 0: T % 4 == 0, launcher) -- one 8-byte half of a 16-byte fragment; the other half (keys t + 8 ..) is the
		// lane's group g + 1 or g - 1 and follows within a few instructions, so the L2 sees whole lines. Per lane 16 offsets (4 row tiles x 4 groups),
		// computed once per tile; the dimension block (+ 1 KiB) is an immediate, the head a scalar base. Same values as epilogueBlockV32x64.
		__device__ __forceinline__ void epilogueFastV4( const GemmArgs& a, f32x16 ( &acc )[ 4 ][ 4 ], int mW, int nW, int lane )
		{
			asm volatile( "" : "+v"( lane ) );
			const int hi = lane >> 5, cl = lane & 31;
			const int d = a.H * HEAD_DIM;
			float bias[ 4 ];
	#pragma unroll
			for( int j = 0; j < 4; j++ ) bias[ j ] = a.bias ? a.bias[ nW + 32 * j + cl ] : 0.0f;
			const int b = __builtin_amdgcn_readfirstlane( mW / a.T );
			const int segPos = mW - b * a.T;
			const unsigned headBytes = (unsigned)HEAD_DIM * (unsigned)a.Tpad * 2u;
			const unsigned seqBytes = (unsigned)a.H * headBytes;
			unsigned char* const base = (unsigned char*)a.v + (long long)b * seqBytes + (long long)( ( nW - 2 * d ) >> 6 ) * headBytes;
			unsigned voff[ 4 ][ 4 ];
	#pragma unroll
			for( int i = 0; i < 4; i++ )
	#pragma unroll
				for( int g = 0; g < 4; g++ )
				{
					int t = segPos + 32 * i + 8 * g + 4 * hi;
					const bool cross = t >= a.T;
					t = cross ? t - a.T : t;
					voff[ i ][ g ] = ( cross ? seqBytes : 0u ) + (unsigned)( ( ( ( t >> 4 ) * 128 + ( ( t >> 2 ) & 1 ) * 32 + cl ) * 8 + ( ( t >> 3 ) & 1 ) * 4 ) * 2 );
				}
	#pragma unroll
			for( int i = 0; i < 4; i++ )
			{
	#pragma unroll
				for( int j = 0; j < 4; j++ )
				{
					unsigned char* const bj = base + ( j >> 1 ) * (long long)headBytes + ( j & 1 ) * 1024;
	#pragma unroll
					for( int g = 0; g < 4; g++ )
					{
						f16x4 pk;
	#pragma unroll
						for( int e = 0; e < 4; e++ )
						{
							float x;
							asm volatile( "v_accvgpr_read_b32 %0, %1" : "=v"( x ) : "a"( acc[ i ][ j ][ 4 * g + e ] ) );
							pk[ e ] = (f16)( x + bias[ j ] );
						}
						*(f16x4*)( bj + voff[ i ][ g ] ) = pk;
					}
				}
				__builtin_amdgcn_sched_barrier( 0 );
			}
		}

		struct Cfg4
		{
			static constexpr int BM = 256, BN = 256, BK = 64, NT = 256, TI = 4, TJ = 4;
			static constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = ( BM + BN ) * BK * 2;	   // 32 KiB, 64 KiB
			static constexpr int EPI_OFFSET = 2 * STAGE_BYTES;
			static constexpr int EPI_PER_WAVE = 4096;
			static constexpr int LDS_BYTES = EPI_OFFSET + 4 * EPI_PER_WAVE;
		};

		// SCH (probe builds; 0 = the instance that ships; all give correct results): 1 = the DMA pieces of a K tile spread 3 / 3 / 2 over three
		// substeps (else 4 / 4 over two), 2 = the compiler's own order inside a chunk, 4 = 2 fragment reads per chunk instead of 4 + 4 + 0 + 0,
		// 16384 = no early W pieces / counted wait after the epilogue
		template<int EPI, bool WIDE, int SCH = 0>
		__global__ void __launch_bounds__( 256, 1 ) gemmTiled4( const GemmArgs a )
		{
			using C = Cfg4;
			constexpr int BM = C::BM, BN = C::BN, BK = C::BK;
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smem[];
			typedef __attribute__( ( address_space( 3 ) ) ) void* LdsPtr;

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
			const int wr = wave >> 1, wc = wave & 1;

			// ---- this workgroup's tiles (as gemmTiled8): XCD x = workgroup id % 8 owns a contiguous range of the band-walk order
			const int tilesM = ( a.M + BM - 1 ) / BM, tilesN = ( a.N + BN - 1 ) / BN;
			const int nTiles = tilesM * tilesN;
			int linFirst, linEnd, linStep;
			{
				const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
				const int q = nTiles >> 3, r = nTiles & 7;
				const int start = xcd < r ? xcd * ( q + 1 ) : r * ( q + 1 ) + ( xcd - r ) * q;
				linEnd = start + ( xcd < r ? q + 1 : q );
				linFirst = start + idx;
				linStep = ( gridDim.x + 7 - xcd ) >> 3;
			}
			auto tileCoords = [ & ]( int lin, int& tm, int& tn )
			{
				if( a.groupM > 1 )
				{
					const int perBand = a.groupM * tilesN;
					const int band = lin / perBand;
					const int first = band * a.groupM;
					const int rows = min( tilesM - first, a.groupM );
					const int r = lin - band * perBand;
					tm = first + r % rows;
					tn = r / rows;
				}
				else
				{
					tm = lin / tilesN;
					tn = lin - tm * tilesN;
				}
			};
			if( linFirst >= linEnd ) return;

			// ---- producer side: LDS-DMA sources. A tile is 32 pieces of 8 rows x 128 bytes; wave w owns pieces 8 w .. 8 w + 7 of the A
			// tile and of the W tile, issued as 4 + 4 pairs. Lane l of a piece lands at row l / 8, physical chunk l % 8, which must hold
			// logical chunk (l % 8) ^ ((row >> 1) & 7); offA / offW = byte offset of that chunk from a.A / a.W at k = 0.
			// pOff* = the output tile the producer is in (recomputed, branch-free, when it moves on to the workgroup's next tile in the middle of the consumer's K loop)
			unsigned pOffA[ 4 ][ 2 ], pOffW[ 4 ][ 2 ];
			auto tileOffsets = [ & ]( int lin, unsigned( &offA )[ 4 ][ 2 ], unsigned( &offW )[ 4 ][ 2 ] )
			{
				int tm, tn;
				tileCoords( lin, tm, tn );
				// No branch and no division per row: rows past M / N repeat the last one, a tile crosses at most one segment boundary of A
				// (segments of at least 256 rows, launcher), and everything fits 32 bits (launcher)
				const int mFirst = tm * BM, nFirst = tn * BN;
				const int mMax = a.M - 1 - mFirst, nMax = a.N - 1 - nFirst;
				int laneV = lane;
				asm volatile( "" : "+v"( laneV ) );	   // (not hoisted out of the tile loop into scratch)
				const int rIn = laneV >> 3, cPhys = laneV & 7;
				const bool segd = a.Mb > 0 && a.Mb < a.M;
				const int b0 = segd ? mFirst / a.Mb : 0;
				const int t0 = mFirst - b0 * ( segd ? a.Mb : 0 );
				const int segLeft = segd ? a.Mb - t0 : 0x7fffffff;
				const unsigned aBase = (unsigned)( ( (long long)b0 * a.aBatchStride + (long long)t0 * a.lda ) * 2 );
				const unsigned crossA = segd ? (unsigned)( ( a.aBatchStride - (long long)a.Mb * a.lda ) * 2 ) : 0u;
				const unsigned wBase = (unsigned)( (long long)nFirst * a.K * 2 );
	#pragma unroll
				for( int q = 0; q < 4; q++ )
	#pragma unroll
					for( int i = 0; i < 2; i++ )
					{
						const int row = ( wave * 8 + q * 2 + i ) * 8 + rIn;
						const unsigned c16 = (unsigned)( cPhys ^ ( ( row >> 1 ) & 7 ) ) * 16u;
						const int rm = min( row, mMax );
						offA[ q ][ i ] = aBase + (unsigned)rm * (unsigned)( a.lda * 2 ) + ( rm >= segLeft ? crossA : 0u ) + c16;
						const int rn = min( row, nMax );
						offW[ q ][ i ] = wBase + (unsigned)rn * (unsigned)( a.K * 2 ) + c16;
					}
			};
			const unsigned ldsBase = __builtin_amdgcn_readfirstlane( (unsigned)(size_t)(LdsPtr)smem );
			const unsigned pieceBase = ldsBase + (unsigned)wave * 8192u;
			const int nk = a.K / BK;	  // >= 2 (launcher)
			int pKt = 0, pLin = linFirst;
			unsigned pBufOff = 0;	  // byte offset of the buffer the producer's K tile goes to
			auto dmaA = [ & ]( auto qc )
			{
				constexpr int q = decltype( qc )::value;
				ldsDmaPair( a.A + pKt * BK, pOffA[ q ][ 0 ], pOffA[ q ][ 1 ], pieceBase + pBufOff + q * 2048 );
			};
			auto dmaW = [ & ]( auto qc )
			{
				constexpr int q = decltype( qc )::value;
				ldsDmaPair( a.W + pKt * BK, pOffW[ q ][ 0 ], pOffW[ q ][ 1 ], pieceBase + pBufOff + C::A_BYTES + q * 2048 );
			};
			using Q0 = std::integral_constant<int, 0>;
			using Q1 = std::integral_constant<int, 1>;
			using Q2 = std::integral_constant<int, 2>;
			using Q3 = std::integral_constant<int, 3>;
			// the producer's next K tile: the one after in this output tile, or K tile 0 of the workgroup's next output tile. Past the
			// workgroup's last tile it keeps issuing (valid addresses of an earlier tile, buffers nobody reads): no branch in the K loop
			auto advanceProducer = [ & ]( unsigned bufOff )
			{
				pBufOff = bufOff;
				if( ++pKt < nk ) return;
				pKt = 0;
				pLin += linStep;
				if( pLin < linEnd ) tileOffsets( pLin, pOffA, pOffW );
			};
			// Which of a K tile's 8 pairs (0..3 = A, 4..7 = W; A first: its rows are the ones that may come from HBM) goes out after chunk c
			// of substep s (s = 3: the last substep of K tile g - 2, s = 0 / 1: the first two of g - 1); -1 = none
			// pos: 0 = a K tile in the middle of an output tile, 1 = the FIRST one (its W pieces went out before the epilogue: nothing in substep 0),
			// 2 = the LAST one (substep 3 issues the next K tile's A AND W pieces: everything the first barrier after the epilogue waits for is then older
			// than the epilogue's stores, and the wait can leave those in flight)
			auto dmaAfter = [ & ]( auto sc, auto cc, auto posc )
			{
				constexpr int s = decltype( sc )::value, c = decltype( cc )::value, pos = decltype( posc )::value;
				if constexpr( ( SCH & 1 ) == 0 && ( SCH & 16384 ) == 0 )
				{
					if constexpr( pos == 1 && s == 0 ) return;
					if constexpr( pos == 2 && s == 3 )
					{
						dmaA( cc );
						dmaW( cc );
						return;
					}
				}
				constexpr int pair = ( SCH & 1 ) == 0 ? ( s == 3 ? c : s == 0 ? 4 + c : -1 )
													  : ( s == 3 ? ( c < 3 ? c : -1 ) : s == 0 ? ( c < 3 ? 3 + c : -1 ) : s == 1 ? ( c < 2 ? 6 + c : -1 ) : -1 );
				if constexpr( pair >= 4 )
					dmaW( std::integral_constant<int, ( pair >= 4 ? pair - 4 : 0 )>{} );
				else if constexpr( pair >= 0 )
					dmaA( std::integral_constant<int, ( pair >= 0 && pair < 4 ? pair : 0 )>{} );
			};

			// ---- consumer side: lane l reads row l & 31 of a 32-row tile, logical chunk 2 ks + (l >> 5), stored at chunk ^ ((row >> 1) & 7)
			const int x0 = ( lane >> 5 ) ^ ( ( lane >> 1 ) & 7 );
			unsigned aAddr[ 4 ], wAddr[ 4 ];	 // byte offsets inside a K-tile buffer, per k-substep
	#pragma unroll
			for( int ks = 0; ks < 4; ks++ )
			{
				const unsigned laneK = (unsigned)( ( lane & 31 ) * 128 + ( ( x0 ^ ( ks << 1 ) ) << 4 ) );
				aAddr[ ks ] = (unsigned)( wr * 128 * 128 ) + laneK;
				wAddr[ ks ] = (unsigned)( C::A_BYTES + wc * 128 * 128 ) + laneK;
			}
			f32x16 acc[ 4 ][ 4 ];
	#pragma unroll
			for( int i = 0; i < 4; i++ )
	#pragma unroll
				for( int j = 0; j < 4; j++ )
	#pragma unroll
					for( int r = 0; r < 16; r++ ) acc[ i ][ j ][ r ] = 0.0f;
			f16x8 fa[ 2 ][ 4 ], fb[ 2 ][ 4 ];
			// One substep (index s of its K tile) = four chunks of 4 MFMAs (A row tile c x the four W tiles) from register set SET; the
			// fragments of the NEXT substep (k-substep ksNext of the buffer at bufOff) go to set SET ^ 1: the W fragments with chunk 0,
			// the A fragments with chunk 1, so that every read has at least 8 MFMAs (256 matrix-pipe cycles) to come back. Nothing
			// crosses a chunk boundary (sched_barrier): a DMA pair issued there sits between two groups of MFMAs in the stream.
			auto substep = [ & ]( auto sc, auto setc, auto zeroc, auto posc, unsigned bufOff, int ksNext )
			{
				constexpr int SET = decltype( setc )::value;
				constexpr bool ZERO = decltype( zeroc )::value;
				const unsigned char* const pa = smem + bufOff + aAddr[ ksNext ];
				const unsigned char* const pw = smem + bufOff + wAddr[ ksNext ];
				auto chunk = [ & ]( auto cc )
				{
					constexpr int c = decltype( cc )::value;
					constexpr int RD = ( SCH & 4 ) ? 2 : 4;	   // reads per chunk: 4 + 4 + 0 + 0 or 2 + 2 + 2 + 2
					if constexpr( ( SCH & 4 ) == 0 )
					{
						if constexpr( c == 0 )
						{
	#pragma unroll
							for( int j = 0; j < 4; j++ ) fb[ SET ^ 1 ][ j ] = *(const f16x8*)( pw + j * 4096 );
						}
						if constexpr( c == 1 )
						{
	#pragma unroll
							for( int i = 0; i < 4; i++ ) fa[ SET ^ 1 ][ i ] = *(const f16x8*)( pa + i * 4096 );
						}
					}
					else
					{
						if constexpr( c < 2 )
						{
	#pragma unroll
							for( int j = 0; j < 2; j++ ) fb[ SET ^ 1 ][ 2 * c + j ] = *(const f16x8*)( pw + ( 2 * c + j ) * 4096 );
						}
						else
						{
	#pragma unroll
							for( int i = 0; i < 2; i++ ) fa[ SET ^ 1 ][ 2 * ( c - 2 ) + i ] = *(const f16x8*)( pa + ( 2 * ( c - 2 ) + i ) * 4096 );
						}
					}
					auto mfmaOne = [ & ]( int j )
					{
						if constexpr( ZERO )
						{
							const f32x16 z = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
							acc[ c ][ j ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( fa[ SET ][ c ], fb[ SET ][ j ], z, 0, 0, 0 );
						}
						else
							acc[ c ][ j ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( fa[ SET ][ c ], fb[ SET ][ j ], acc[ c ][ j ], 0, 0, 0 );
					};
	#pragma unroll
					for( int j = 0; j < 4; j++ ) mfmaOne( j );
					if constexpr( ( SCH & 2 ) == 0 && ( ( SCH & 4 ) != 0 || c < 2 ) )
					{
						// MFMA first, then a read behind each MFMA
	#pragma unroll
						for( int k = 0; k < RD; k++ )
						{
							__builtin_amdgcn_sched_group_barrier( 0x008, 1, 0 );
							__builtin_amdgcn_sched_group_barrier( 0x100, 1, 0 );
						}
						if constexpr( RD < 4 ) __builtin_amdgcn_sched_group_barrier( 0x008, 4 - RD, 0 );
					}
					__builtin_amdgcn_sched_barrier( 0 );
					dmaAfter( sc, cc, posc );
					__builtin_amdgcn_sched_barrier( 0 );
				};
				chunk( std::integral_constant<int, 0>{} );
				chunk( std::integral_constant<int, 1>{} );
				chunk( std::integral_constant<int, 2>{} );
				chunk( std::integral_constant<int, 3>{} );
			};
			using S0 = std::integral_constant<int, 0>;
			using S1 = std::integral_constant<int, 1>;
			using P0 = std::integral_constant<int, 0>;
			using P1 = std::integral_constant<int, 1>;
			using P2 = std::integral_constant<int, 2>;
			using P3 = std::integral_constant<int, 3>;
			using ZN = std::integral_constant<bool, false>;
			using ZY = std::integral_constant<bool, true>;

			unsigned char* const stage = smem + C::EPI_OFFSET + wave * C::EPI_PER_WAVE;

			unsigned bufOff = 0;
			// One K tile of the consumer; the fragments of its first substep are in register set 0. There is ONE instance of every K tile position
			// (first / middle / last) in a row, never alternatives: accumulators that meet at the end of alternative paths are 256 registers
			// the allocator then copies around.
			int postEpi = 0;	 // VMEM operations the last epilogue issued after the DMA pieces of the K tile that follows it (0 / 32 / 64: see the wait below)
			auto kTile = [ & ]( auto zeroc, auto posc )
			{
				constexpr int pos = decltype( posc )::value;
				substep( P0{}, S0{}, zeroc, posc, bufOff, 1 );
				substep( P1{}, S1{}, ZN{}, posc, bufOff, 2 );
				// substep 2; then every fragment of this buffer is in registers and this wave's pieces of the next K tile must have landed
				substep( P2{}, S0{}, ZN{}, posc, bufOff, 3 );
				if( pos == 1 && ( SCH & 16384 ) == 0 && postEpi >= 63 )
					// the first K tile after an epilogue: its successor's pieces are all OLDER than the epilogue's loads and stores (vmcnt is one
					// in-order queue), so they have landed as soon as no more than those are in flight -- the stores go on draining under this
					// K tile and the next
					asm volatile( "s_waitcnt vmcnt(63) lgkmcnt(0)" ::: "memory" );
				else if( pos == 1 && ( SCH & 16384 ) == 0 && postEpi >= 32 )
					asm volatile( "s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory" );
				else
					asm volatile( "s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory" );
				WH_BAR();
				// substep 3: the next K tile is complete in the other buffer, this buffer is dead
				advanceProducer( bufOff );
				bufOff ^= (unsigned)C::STAGE_BYTES;
				substep( P3{}, S1{}, ZN{}, posc, bufOff, 0 );
			};

			auto epilogue = [ & ]( int tmD, int tnD, bool lastTile )
			{
				postEpi = 0;
				if constexpr( WIDE )
				{
					const int mW = tmD * BM + wr * 128, nW = tnD * BN + wc * 128;
					bool isV = false;
					if constexpr( EPI == EPI_QKV_ENC ) isV = nW >= 2 * a.H * HEAD_DIM;	   // 2 d is a multiple of 256: a tile is V or it is not
					const bool interior = a.wideEpi == 2 && ( tmD + 1 ) * BM <= a.M && ( tnD + 1 ) * BN <= a.N;
					if constexpr( EPI == EPI_QKV_ENC )
					{
						if( isV && interior )
						{
							epilogueFastV4( a, acc, mW, nW, lane );
							postEpi = 64;
							return;
						}
					}
					if( !isV && interior )
					{
						if constexpr( EPI == EPI_F32 )
						{
							if( a.res )
								epilogueFast4<EPI, true>( a, acc, mW, nW, lane, stage );
							else
								epilogueFast4<EPI, false>( a, acc, mW, nW, lane, stage );
						}
						else
							epilogueFast4<EPI, false>( a, acc, mW, nW, lane, stage );
						postEpi = EPI == EPI_F32 ? 64 : 32;
						return;
					}
					// edge tiles (and launches without the fast path's promises): the general block epilogues of gemmTiled8
					int laneS = lane;
					asm volatile( "" : "+v"( laneS ) );
	#pragma unroll
					for( int i = 0; i < 4; i++ )
	#pragma unroll
						for( int jp = 0; jp < 2; jp++ )
						{
							const int m0 = mW + i * 32, n0 = nW + jp * 64;
							if constexpr( EPI == EPI_QKV_ENC )
							{
								// fragment-major V straight from the registers (the launcher guarantees T % 4 == 0 for this instance)
								if( isV )
								{
									const f32x16 c0 = accReadTile( acc[ i ][ 2 * jp ] ), c1 = accReadTile( acc[ i ][ 2 * jp + 1 ] );
									epilogueBlockV32x64( a, c0, c1, m0, n0, laneS );
									__builtin_amdgcn_sched_barrier( 0 );
									continue;
								}
							}
							const f32x16 c0 = accReadTile( acc[ i ][ 2 * jp ] ), c1 = accReadTile( acc[ i ][ 2 * jp + 1 ] );
							epilogueBlock32x64<EPI>( a, c0, c1, m0, n0, laneS, stage );
							__builtin_amdgcn_sched_barrier( 0 );
						}
				}
				else
				{
					// element-wise stores (N % 8 != 0 and the like): a copy of the tile in VGPRs, most of it through scratch -- correct, not fast
					f32x16 cp[ 4 ][ 4 ];
	#pragma unroll
					for( int i = 0; i < 4; i++ )
	#pragma unroll
						for( int j = 0; j < 4; j++ ) cp[ i ][ j ] = accReadTile( acc[ i ][ j ] );
					tileEpilogue<EPI, Cfg4>( a, cp, tmD, tnD, wr, wc, lane );
				}
			};

			// ---- prologue: K tile 0 of the first output tile completely, then the first part of K tile 1
			int lin = linFirst;
			tileOffsets( lin, pOffA, pOffW );
			dmaA( Q0{} );
			dmaA( Q1{} );
			dmaA( Q2{} );
			dmaA( Q3{} );
			dmaW( Q0{} );
			dmaW( Q1{} );
			dmaW( Q2{} );
			dmaW( Q3{} );
			asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
			WH_BAR();
			advanceProducer( C::STAGE_BYTES );
			dmaA( Q0{} );
			dmaA( Q1{} );
			dmaA( Q2{} );
			if constexpr( ( SCH & 1 ) == 0 ) dmaA( Q3{} );
			if constexpr( ( SCH & 1 ) == 0 && ( SCH & 16384 ) == 0 )
			{
				dmaW( Q0{} );
				dmaW( Q1{} );
				dmaW( Q2{} );
				dmaW( Q3{} );
			}
	#pragma unroll
			for( int i = 0; i < 4; i++ ) fa[ 0 ][ i ] = *(const f16x8*)( smem + aAddr[ 0 ] + i * 4096 );
	#pragma unroll
			for( int j = 0; j < 4; j++ ) fb[ 0 ][ j ] = *(const f16x8*)( smem + wAddr[ 0 ] + j * 4096 );
			__builtin_amdgcn_sched_barrier( 0 );

			using KM = std::integral_constant<int, 0>;
			using KF = std::integral_constant<int, 1>;
			using KL = std::integral_constant<int, 2>;
			for( ;; )
			{
				// first, middle, last (nk >= 2)
				kTile( ZY{}, KF{} );
				for( int kt = 1; kt + 1 < nk; kt++ ) kTile( ZN{}, KM{} );
				kTile( ZN{}, KL{} );
				int tm, tn;
				tileCoords( lin, tm, tn );
				asm volatile( "s_nop 15\n\ts_nop 15" ::: "memory" );	   // the last MFMA's 16 passes are over before the first accumulator is read
				epilogue( tm, tn, lin + linStep >= linEnd );
				lin += linStep;
				if( lin >= linEnd ) break;
			}
			// the producer ran ahead: nothing of it may land after the workgroup has given its LDS back
			asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
		}
#undef WH_BAR

		// ---- skinny: M <= 32 ----
		constexpr int SK_WAVES = 4;

		template<int EPI>
		__global__ void __launch_bounds__( 256 ) gemmSkinny( const GemmArgs a )
		{
			__shared__ float red[ SK_WAVES - 1 ][ 16 ][ 64 ];

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = tid >> 6;
			const int n0 = blockIdx.x * 32;

			int n = n0 + ( lane & 31 );
			n = n < a.N ? n : a.N - 1;
			int m = lane & 31;
			m = m < a.M ? m : a.M - 1;
			const int kPer = a.K / SK_WAVES;
			const int kBeg = wave * kPer + ( lane >> 5 ) * 8;
			const f16* pw = a.W + (long long)n * a.K + kBeg;
			const f16* px = a.A + rowOffset( m, a.Mb, a.lda, a.aBatchStride ) + kBeg;

			f32x16 acc;
#pragma unroll
			for( int r = 0; r < 16; r++ ) acc[ r ] = 0.0f;

			const int steps = kPer / 16;
			int s = 0;
			for( ; s + 4 <= steps; s += 4 )
			{
				f16x8 fw[ 4 ], fx[ 4 ];
#pragma unroll
				for( int u = 0; u < 4; u++ )
				{
					fw[ u ] = __builtin_nontemporal_load( (const f16x8*)( pw + ( s + u ) * 16 ) );
					fx[ u ] = *(const f16x8*)( px + ( s + u ) * 16 );
				}
#pragma unroll
				for( int u = 0; u < 4; u++ )
					acc = __builtin_amdgcn_mfma_f32_32x32x16_f16( fw[ u ], fx[ u ], acc, 0, 0, 0 );
			}
			for( ; s < steps; s++ )
			{
				const f16x8 fw = *(const f16x8*)( pw + s * 16 );
				const f16x8 fx = *(const f16x8*)( px + s * 16 );
				acc = __builtin_amdgcn_mfma_f32_32x32x16_f16( fw, fx, acc, 0, 0, 0 );
			}

			if( wave > 0 )
			{
#pragma unroll
				for( int r = 0; r < 16; r++ ) red[ wave - 1 ][ r ][ lane ] = acc[ r ];
			}
			__syncthreads();
			if( wave != 0 ) return;
#pragma unroll
			for( int w = 0; w < SK_WAVES - 1; w++ )
#pragma unroll
				for( int r = 0; r < 16; r++ ) acc[ r ] += red[ w ][ r ][ lane ];

			// D[row][col]: row = weight row (n), col = activation row (m)
			const int mm = lane & 31;
			if( mm >= a.M ) return;
			const int hi = lane >> 5;
#pragma unroll
			for( int r = 0; r < 16; r++ )
			{
				const int nn = n0 + ( r & 3 ) + 8 * ( r >> 2 ) + 4 * hi;
				if( nn < a.N )
					epilogueOne<EPI>( a, mm, nn, acc[ r ] );
			}
		}
		// ---- gemv: M <= 32 activation rows (single-token decode steps of a lock-step batch; MT = 2 above 16 rows) ----
		// HBM/latency-bound: the only thing that matters is how many weight bytes are in flight. 16 weight rows per
		// workgroup (N/16 workgroups), 4 waves split K, and every wave issues ALL of its weight loads (16 bytes per lane
		// each, up to GV_UNROLL (8 or 16) at a time) before the first MFMA consumes one. v_mfma_f32_16x16x32_f16: A = 16 weight rows,
		// B = up to 16 activation rows. With lnX != null the LayerNorm that precedes the product in the graph
		// (norm.hlsl + fmaRepeat1.hlsl in the reference) runs as a prologue: each workgroup normalises the M rows into LDS
		// (FP16, the rounding the product applies anyway) -- M*K*4 bytes of L2 reads per workgroup instead of a launch.
		constexpr int GV_UNROLL_MAX = 16;
		constexpr int GV_MAXK_LN = 1280;
		constexpr int GV_XS_STRIDE = GV_MAXK_LN + 8;

		// LayerNorm + affine of up to RB rows by the WHOLE workgroup (NWV waves): thread t owns the float4 columns t and
		// t + 64 * NWV of every row, so a row is one coalesced pass and all RB rows are in flight at once; the two reductions go
		// wave-shuffle -> LDS -> every thread. Same formula as layerNormRows (two-pass FP32, eps 1e-5, w*y + b, FP16 result);
		// the summation tree differs, so rows are not bit-identical with the one-wave-per-row version.
		template<int RB, int MAXC, int NWV, class Store>
		__device__ __forceinline__ void layerNormBlock( const float* __restrict__ x, int nRows, const float* __restrict__ w, const float* __restrict__ b,
			int d, int tid, float ( *shA )[ RB ], float ( *shB )[ RB ], Store&& store )
		{
			constexpr int NTH = NWV * 64;
			const int lane = tid & 63, wave = tid >> 6;
			const int nv = d >> 2;
			f32x4 v[ RB ][ MAXC ], wv[ MAXC ], bv[ MAXC ];
	#pragma unroll
			for( int i = 0; i < MAXC; i++ )
			{
				const int cv = tid + i * NTH;
				const int cc = ( cv < nv ? cv : nv - 1 ) * 4;
				wv[ i ] = *(const f32x4*)( w + cc );
				bv[ i ] = *(const f32x4*)( b + cc );
	#pragma unroll
				for( int r = 0; r < RB; r++ )
				{
					const int rr = r < nRows ? r : ( nRows > 0 ? nRows - 1 : 0 );
					v[ r ][ i ] = *(const f32x4*)( x + (long long)rr * d + cc );
				}
			}
			const float invD = 1.0f / (float)d;
			float s[ RB ];
	#pragma unroll
			for( int r = 0; r < RB; r++ )
			{
				float t = 0.0f;
	#pragma unroll
				for( int i = 0; i < MAXC; i++ )
					if( tid + i * NTH < nv ) t += ( v[ r ][ i ][ 0 ] + v[ r ][ i ][ 1 ] ) + ( v[ r ][ i ][ 2 ] + v[ r ][ i ][ 3 ] );
				s[ r ] = t;
			}
	#pragma unroll
			for( int o = 32; o > 0; o >>= 1 )
	#pragma unroll
				for( int r = 0; r < RB; r++ ) s[ r ] += __shfl_xor( s[ r ], o, 64 );
			if( lane == 0 )
	#pragma unroll
				for( int r = 0; r < RB; r++ ) shA[ wave ][ r ] = s[ r ];
			__syncthreads();
	#pragma unroll
			for( int r = 0; r < RB; r++ )
			{
				float t = shA[ 0 ][ r ];
	#pragma unroll
				for( int ww = 1; ww < NWV; ww++ ) t += shA[ ww ][ r ];
				const float mean = t * invD;
				float q = 0.0f;
	#pragma unroll
				for( int i = 0; i < MAXC; i++ )
				{
	#pragma unroll
					for( int e = 0; e < 4; e++ ) v[ r ][ i ][ e ] -= mean;
					if( tid + i * NTH < nv )
	#pragma unroll
						for( int e = 0; e < 4; e++ ) q = fmaf( v[ r ][ i ][ e ], v[ r ][ i ][ e ], q );
				}
				s[ r ] = q;
			}
	#pragma unroll
			for( int o = 32; o > 0; o >>= 1 )
	#pragma unroll
				for( int r = 0; r < RB; r++ ) s[ r ] += __shfl_xor( s[ r ], o, 64 );
			if( lane == 0 )
	#pragma unroll
				for( int r = 0; r < RB; r++ ) shB[ wave ][ r ] = s[ r ];
			__syncthreads();
	#pragma unroll
			for( int r = 0; r < RB; r++ )
			{
				if( r >= nRows ) continue;
				float t = shB[ 0 ][ r ];
	#pragma unroll
				for( int ww = 1; ww < NWV; ww++ ) t += shB[ ww ][ r ];
				const float rstd = 1.0f / sqrtf( t * invD + 1e-5f );
	#pragma unroll
				for( int i = 0; i < MAXC; i++ )
				{
					const int cv = tid + i * NTH;
					if( cv < nv )
					{
						f16x4 hv;
	#pragma unroll
						for( int e = 0; e < 4; e++ ) hv[ e ] = (f16)__fadd_rn( __fmul_rn( __fmul_rn( v[ r ][ i ][ e ], rstd ), wv[ i ][ e ] ), bv[ i ][ e ] );
						store( r, cv * 4, hv );
					}
				}
			}
		}

		// PRO = 0: A rows are FP16 in global memory; 1: fused LayerNorm prologue, a wave per pair of rows (up to 16 rows);
		// 2: fused LayerNorm prologue by the whole workgroup, 16 rows at a time (17 .. 32 rows).
		// ROWS = weight rows per workgroup: 16 fills the MFMA; 4 (rows replicated across the operand's 16 row slots) gives 4x
		// the workgroups when N is small and K large -- a CU streams only ~24 GB/s, so 8 MB over 64 CUs would take 5 us.
		// NW = waves per workgroup that split K. GV_UNROLL = fragment slots per wave (8 halves the registers when K / NW / 32 <= 8).
		// MT = MFMA column tiles = 16 activation rows each.
		template<int EPI, int PRO, int ROWS, int NW, int GV_UNROLL, int MT>
		__global__ void __launch_bounds__( NW * 64 ) gemvFused( const GemmArgs a )
		{
			constexpr bool LN = PRO == 1;
			__shared__ float red[ NW - 1 ][ MT * 4 ][ 64 ];
			__shared__ float lnA[ PRO == 2 ? NW : 1 ][ 16 ], lnB2[ PRO == 2 ? NW : 1 ][ 16 ];
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) f16 xs[];	 // [16 * MT][GV_XS_STRIDE] when there is a prologue

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = tid >> 6;
			const int n0 = blockIdx.x * ROWS;
			// more than 16 * MT activation rows: blockIdx.y selects the group of 16 * MT rows (the weight rows are streamed once
			// per group; these launches are latency-bound, the second copy comes from L2 or overlaps the first)
			const int m0 = blockIdx.y * 16 * MT;
			const int mEnd = a.M;

			int n = n0 + ( lane & 15 ) % ROWS;
			n = n < a.N ? n : a.N - 1;
			const int kPer = a.K / NW;
			const int kBeg = wave * kPer + ( lane >> 4 ) * 8;
			const f16* const pw = a.W + (long long)n * a.K + kBeg;
			const int steps = kPer / 32;

			// first batch of weight loads goes out before anything else: it does not depend on the LayerNorm prologue
			f16x8 fw[ GV_UNROLL ], fx[ MT ][ GV_UNROLL ];
#pragma unroll
			for( int u = 0; u < GV_UNROLL; u++ )
				if( u < steps ) fw[ u ] = __builtin_nontemporal_load( (const f16x8*)( pw + u * 32 ) );

			// epilogue operands of the plain FP32 epilogue are fetched up front as well (wave 0 owns the epilogue)
			const int nEp = n0 + ( lane >> 4 ) * 4;
			const bool fastEp = EPI == EPI_F32 && ( a.N & 15 ) == 0 && a.Mb >= a.M;
			const bool ownsRows = ( lane >> 4 ) * 4 < ROWS;	  // with ROWS == 4 only the first 16 lanes hold distinct output rows
			f32x4 biasv = { 0.0f, 0.0f, 0.0f, 0.0f }, resv[ MT ];
#pragma unroll
			for( int t = 0; t < MT; t++ ) resv[ t ] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
			if( fastEp && wave == 0 && ownsRows )
			{
				if( a.bias ) biasv = *(const f32x4*)( a.bias + nEp );
#pragma unroll
				for( int t = 0; t < MT; t++ )
					if( a.res && m0 + t * 16 + ( lane & 15 ) < mEnd ) resv[ t ] = *(const f32x4*)( a.res + (long long)( m0 + t * 16 + ( lane & 15 ) ) * a.ldc + nEp );
			}

			const f16* px[ MT ];
			if constexpr( LN )
			{
				// rows wave and wave + NW together, then the next pair. Rows at or beyond M stay unwritten: an MFMA output
				// column depends on its own activation row only, and those columns are never stored.
				for( int r0 = wave; r0 < a.M; r0 += 2 * NW )
				{
					const int nr = ( a.M - r0 + NW - 1 ) / NW;
					layerNormRows<GV_MAXK_LN / 256, 2>( a.lnX + (long long)r0 * a.K, (long long)NW * a.K, nr, a.lnW, a.lnB, a.K, lane,
						[ = ]( int j, int c, f16x4 v ) { *(f16x4*)( xs + ( r0 + j * NW ) * GV_XS_STRIDE + c ) = v; } );
				}
				__syncthreads();
			}
			if constexpr( PRO == 2 )
			{
				for( int r0 = 0; r0 < a.M; r0 += 16 )
					layerNormBlock<16, ( GV_MAXK_LN / 4 + NW * 64 - 1 ) / ( NW * 64 ), NW>( a.lnX + (long long)r0 * a.K, a.M - r0, a.lnW, a.lnB, a.K, tid, lnA, lnB2,
						[ = ]( int j, int c, f16x4 v ) { *(f16x4*)( xs + ( r0 + j ) * GV_XS_STRIDE + c ) = v; } );
				__syncthreads();
			}
#pragma unroll
			for( int t = 0; t < MT; t++ )
			{
				if constexpr( PRO != 0 )
					px[ t ] = xs + ( t * 16 + ( lane & 15 ) ) * GV_XS_STRIDE + kBeg;
				else
				{
					int m = m0 + t * 16 + ( lane & 15 );
					m = m < mEnd ? m : mEnd - 1;
					px[ t ] = a.A + rowOffset( m, a.Mb, a.lda, a.aBatchStride ) + kBeg;
				}
			}

			f32x4 acc[ MT ];
#pragma unroll
			for( int t = 0; t < MT; t++ ) acc[ t ] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
			for( int u = 0; u < GV_UNROLL; u++ )
				if( u < steps )
				{
#pragma unroll
					for( int t = 0; t < MT; t++ ) fx[ t ][ u ] = *(const f16x8*)( px[ t ] + u * 32 );
				}
#pragma unroll
			for( int u = 0; u < GV_UNROLL; u++ )
				if( u < steps )
				{
#pragma unroll
					for( int t = 0; t < MT; t++ ) acc[ t ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( fw[ u ], fx[ t ][ u ], acc[ t ], 0, 0, 0 );
				}
			for( int s = GV_UNROLL; s < steps; s += GV_UNROLL )
			{
#pragma unroll
				for( int u = 0; u < GV_UNROLL; u++ )
					if( s + u < steps )
					{
						fw[ u ] = __builtin_nontemporal_load( (const f16x8*)( pw + ( s + u ) * 32 ) );
#pragma unroll
						for( int t = 0; t < MT; t++ ) fx[ t ][ u ] = *(const f16x8*)( px[ t ] + ( s + u ) * 32 );
					}
#pragma unroll
				for( int u = 0; u < GV_UNROLL; u++ )
					if( s + u < steps )
					{
#pragma unroll
						for( int t = 0; t < MT; t++ ) acc[ t ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( fw[ u ], fx[ t ][ u ], acc[ t ], 0, 0, 0 );
					}
			}

			if( wave > 0 )
			{
#pragma unroll
				for( int t = 0; t < MT; t++ )
#pragma unroll
					for( int r = 0; r < 4; r++ ) red[ wave - 1 ][ t * 4 + r ][ lane ] = acc[ t ][ r ];
			}
			__syncthreads();
			if( wave != 0 ) return;
#pragma unroll
			for( int w = 0; w < NW - 1; w++ )
#pragma unroll
				for( int t = 0; t < MT; t++ )
#pragma unroll
					for( int r = 0; r < 4; r++ ) acc[ t ][ r ] += red[ w ][ t * 4 + r ][ lane ];

			// D[row][col]: col = lane & 15 = activation row within the tile, row = (lane >> 4) * 4 + r = weight row slot
			if( !ownsRows ) return;
#pragma unroll
			for( int t = 0; t < MT; t++ )
			{
				const int mm = m0 + t * 16 + ( lane & 15 );
				if( mm >= mEnd ) continue;
				if( fastEp )
				{
					// out = (acc + bias) + res, the same order as epilogueOne<EPI_F32>
					f32x4 o;
#pragma unroll
					for( int r = 0; r < 4; r++ ) o[ r ] = ( acc[ t ][ r ] + biasv[ r ] ) + resv[ t ][ r ];
					*(f32x4*)( a.out32 + (long long)mm * a.ldc + nEp ) = o;
					continue;
				}
#pragma unroll
				for( int r = 0; r < 4; r++ )
				{
					const int nn = n0 + ( lane >> 4 ) * 4 + r;
					if( nn < a.N )
						epilogueOne<EPI>( a, mm, nn, acc[ t ][ r ] );
				}
			}
		}

		// -----------------------------------------------------------------------------------------------------------
		// gemmAllRows: 33 .. 128 activation rows against a WIDE weight matrix (the vocabulary projection of a decode step:
		// N = 51865). A workgroup owns 32 columns x ALL rows: the weights are fetched once (gemvFused fetches them once per
		// group of 64 rows) and the activation rows are re-read once per 32 columns instead of once per 16. The 4 waves split
		// K; a wave keeps MT x 2 MFMA 16x16x32 tiles and has two k-steps of loads (2 weight + MT activation fragments each)
		// in flight; the 4 partial tiles meet in LDS and wave w finishes accumulator groups w, w + 4, ... in the fixed order
		// 0, 1, 2, 3. Measured at 112 rows, N = 51865, K = 1024: 112 us (950 GB/s) vs 165 us for gemvFused.
		// It needs N / 32 >= ~500 workgroups to fill the chip. Splitting K over MORE workgroups for the narrow products
		// (N = 1024: 32 column tiles) was built and measured -- partial sums to a scratch buffer, __threadfence, one atomic
		// ticket per tile, last arrival adds the slices in slice order -- and retired: the agent-scope fences (an L2 write-back
		// per workgroup on gfx950) cost 5-30 us per launch, 40-78 us against gemvFused's 8-22 us.
		template<int EPI, int MT>
		__global__ void __launch_bounds__( 256 ) gemmAllRows( const GemmArgs a )
		{
			constexpr int NW = 4, CT = 2, G = MT * CT;
			constexpr int GPW = ( G + NW - 1 ) / NW;	   // accumulator groups (4 registers x 64 lanes) a wave owns after the LDS exchange
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) float redK[];	 // [NW][G * 4][64]

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = tid >> 6;
			const int tile = blockIdx.x;
			const int n0 = tile * 16 * CT;
			const int kPer = a.K / NW;
			const int kBeg = wave * kPer + ( lane >> 4 ) * 8;
			const int steps = kPer / 32;

			const f16* pw[ CT ];
	#pragma unroll
			for( int c = 0; c < CT; c++ )
			{
				int n = n0 + c * 16 + ( lane & 15 );
				n = n < a.N ? n : a.N - 1;
				pw[ c ] = a.W + (long long)n * a.K + kBeg;
			}
			const f16* px[ MT ];
	#pragma unroll
			for( int t = 0; t < MT; t++ )
			{
				int m = t * 16 + ( lane & 15 );
				m = m < a.M ? m : a.M - 1;
				px[ t ] = a.A + rowOffset( m, a.Mb, a.lda, a.aBatchStride ) + kBeg;
			}

			f32x4 acc[ MT ][ CT ];
	#pragma unroll
			for( int t = 0; t < MT; t++ )
	#pragma unroll
				for( int c = 0; c < CT; c++ ) acc[ t ][ c ] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };

			for( int s0 = 0; s0 < steps; s0 += 2 )
			{
				f16x8 fw[ 2 ][ CT ], fx[ 2 ][ MT ];
	#pragma unroll
				for( int u = 0; u < 2; u++ )
					if( s0 + u < steps )
					{
	#pragma unroll
						for( int c = 0; c < CT; c++ ) fw[ u ][ c ] = __builtin_nontemporal_load( (const f16x8*)( pw[ c ] + ( s0 + u ) * 32 ) );
	#pragma unroll
						for( int t = 0; t < MT; t++ ) fx[ u ][ t ] = *(const f16x8*)( px[ t ] + ( s0 + u ) * 32 );
					}
	#pragma unroll
				for( int u = 0; u < 2; u++ )
					if( s0 + u < steps )
					{
	#pragma unroll
						for( int t = 0; t < MT; t++ )
	#pragma unroll
							for( int c = 0; c < CT; c++ )
								acc[ t ][ c ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( fw[ u ][ c ], fx[ u ][ t ], acc[ t ][ c ], 0, 0, 0 );
					}
			}

			// ---- the 4 K-quarters of the workgroup meet in LDS ----
	#pragma unroll
			for( int t = 0; t < MT; t++ )
	#pragma unroll
				for( int c = 0; c < CT; c++ )
	#pragma unroll
					for( int r = 0; r < 4; r++ ) redK[ ( wave * G * 4 + ( t * CT + c ) * 4 + r ) * 64 + lane ] = acc[ t ][ c ][ r ];
			__syncthreads();
			f32x4 part[ GPW ];
	#pragma unroll
			for( int i = 0; i < GPW; i++ )
			{
				const int g = wave + NW * i;
				if( g >= G ) continue;
	#pragma unroll
				for( int r = 0; r < 4; r++ )
				{
					float v = redK[ ( 0 * G * 4 + g * 4 + r ) * 64 + lane ];
	#pragma unroll
					for( int w = 1; w < NW; w++ ) v += redK[ ( w * G * 4 + g * 4 + r ) * 64 + lane ];
					part[ i ][ r ] = v;
				}
			}

			// ---- epilogue: group g = (row tile t, column fragment c); D[row][col]: col = lane & 15 = activation row, row = weight row slot
			const bool fastEp = EPI == EPI_F32 && ( a.N & 3 ) == 0 && a.Mb >= a.M;
	#pragma unroll
			for( int i = 0; i < GPW; i++ )
			{
				const int g = wave + NW * i;
				if( g >= G ) continue;
				const int t = g / CT, c = g - t * CT;
				const int mm = t * 16 + ( lane & 15 );
				const int nn = n0 + c * 16 + ( lane >> 4 ) * 4;
				if( mm >= a.M || nn >= a.N ) continue;
				if( fastEp )
				{
					// out = (acc + bias) + res, the same order as epilogueOne<EPI_F32>
					f32x4 o = part[ i ];
					if( a.bias )
					{
						const f32x4 bv = *(const f32x4*)( a.bias + nn );
	#pragma unroll
						for( int r = 0; r < 4; r++ ) o[ r ] += bv[ r ];
					}
					const long long off = (long long)mm * a.ldc + nn;
					if( a.res )
					{
						const f32x4 rv = *(const f32x4*)( a.res + off );
	#pragma unroll
						for( int r = 0; r < 4; r++ ) o[ r ] += rv[ r ];
					}
					*(f32x4*)( a.out32 + off ) = o;
					continue;
				}
	#pragma unroll
				for( int r = 0; r < 4; r++ )
					if( nn + r < a.N ) epilogueOne<EPI>( a, mm, nn + r, part[ i ][ r ] );
			}
		}

		// -----------------------------------------------------------------------------------------------------------
		// The epilogue of the decode-rows kernels: group g = wave + NW i of the workgroup's MT x CT tiles of 16 x 16 is in part[ i ] -- D[row][col]: col = lane & 15 =
		// activation row of the tile, row = (lane >> 4) * 4 + r = weight row slot (four consecutive output columns of one activation row per lane)
		template<int EPI, int MT, int CT, int NW>
		__device__ __forceinline__ void decRowsEpilogue( const GemmArgs& a, const f32x4 ( &part )[ ( MT * CT + NW - 1 ) / NW ], int m0, int n0, int wave, int lane )
		{
			constexpr int G = MT * CT;
			constexpr int GPW = ( G + NW - 1 ) / NW;
			const bool fast32 = EPI == EPI_F32 && ( a.N & 3 ) == 0 && a.Mb >= a.M;
			const bool fastGelu = EPI == EPI_F16_GELU && ( a.N & 3 ) == 0 && a.Mb >= a.M;
	#pragma unroll
			for( int i = 0; i < GPW; i++ )
			{
				const int g = wave + NW * i;
				if( g >= G ) continue;
				const int t = g / CT, c = g - t * CT;
				const int mm = m0 + t * 16 + ( lane & 15 );
				const int nn = n0 + c * 16 + ( lane >> 4 ) * 4;
				if( mm >= a.M || nn >= a.N ) continue;
				if( fast32 )
				{
					// out = (acc + bias) + res, the same order as epilogueOne<EPI_F32>
					f32x4 o = part[ i ];
					if( a.bias )
					{
						const f32x4 bv = *(const f32x4*)( a.bias + nn );
	#pragma unroll
						for( int r = 0; r < 4; r++ ) o[ r ] += bv[ r ];
					}
					const long long off = (long long)mm * a.ldc + nn;
					if( a.res )
					{
						const f32x4 rv = *(const f32x4*)( a.res + off );
	#pragma unroll
						for( int r = 0; r < 4; r++ ) o[ r ] += rv[ r ];
					}
					*(f32x4*)( a.out32 + off ) = o;
					continue;
				}
				if( fastGelu )
				{
					// gelu16( acc + bias ), the arithmetic of epilogueOne<EPI_F16_GELU>, four columns as one 8-byte store
					const f32x4 bv = *(const f32x4*)( a.bias + nn );
					f16x4 hv;
	#pragma unroll
					for( int r = 0; r < 4; r++ ) hv[ r ] = gelu16( part[ i ][ r ] + bv[ r ] );
					*(f16x4*)( a.out16 + (long long)mm * a.ldc + nn ) = hv;
					continue;
				}
				if constexpr( EPI == EPI_QKV_DEC )
				{
					// the arithmetic of epilogueOne<EPI_QKV_DEC> on the lane's four consecutive columns (one head, one of Q / K / V: d and HEAD_DIM are multiples of 4),
					// leaving as ONE 8-byte store: one pair of divisions and one position load per lane instead of four, a quarter of the store instructions
					if( ( a.N & 3 ) == 0 )
					{
						const int d = a.H * HEAD_DIM;
						const int sel = nn / d;
						const int c = nn - sel * d;
						f16x4 hv;
						f16* dst;
						if( sel == 0 )
						{
							const f32x4 bv = *(const f32x4*)( a.bias + nn );
	#pragma unroll
							for( int r = 0; r < 4; r++ ) hv[ r ] = (f16)( ( part[ i ][ r ] + bv[ r ] ) * a.scale );
							dst = a.q + (long long)mm * d + c;
						}
						else
						{
							const int h = c >> 6, dd = c & 63;
							const int b = mm / a.nTok;
							const int pos = ( a.nPastDev ? a.nPastDev[ b ] : a.nPast ) + ( mm - b * a.nTok );
							const long long o = ( ( (long long)b * a.H + h ) * a.textCtx + pos ) * HEAD_DIM + dd;
							if( sel == 1 )
							{
	#pragma unroll
								for( int r = 0; r < 4; r++ ) hv[ r ] = (f16)( part[ i ][ r ] * a.scale );
								dst = a.k + o;
							}
							else
							{
								const f32x4 bv = *(const f32x4*)( a.bias + nn );
	#pragma unroll
								for( int r = 0; r < 4; r++ ) hv[ r ] = (f16)( part[ i ][ r ] + bv[ r ] );
								dst = a.v + o;
							}
						}
						*(f16x4*)dst = hv;
						continue;
					}
				}
	#pragma unroll
				for( int r = 0; r < 4; r++ )
					if( nn + r < a.N ) epilogueOne<EPI>( a, mm, nn + r, part[ i ][ r ] );
			}
		}

		// gemmDecRows: the products of a decode step whose lock-step batch is LARGER than 128 sequences (129 .. 512 rows: one
		// context of 224 .. 448 windows instead of two of 112). At that many rows a product is a small GEMM (448 x 4096 x 1024:
		// 3.8 GFLOP against 8 MB of weights), and gemvFused's 16-column workgroups would re-read the activation rows once per
		// 16 columns: 64 KB of L2 -> CU traffic per 16 x 64 outputs. Here a workgroup owns 16 CT weight rows x 16 MT activation
		// rows (64 x 64 by default: 8 fragment loads feed 16 MFMAs per k-step and wave, 2.5 x fewer bytes per output), the 4
		// waves split K exactly as gemvFused's do and their partial tiles meet in LDS in wave order 0, 1, 2, 3 -- the same
		// summation order, so a row's result does not depend on which of the two kernels (or which row tile) computed it.
		// Grid (column tiles, row tiles). Operands come straight from L2 (every wave reads its own K quarter: nothing to share
		// through LDS); two k-steps of loads are in flight per wave.
		template<int EPI, int MT, int CT, int DEPTH, int NW = 4>
		__global__ void __launch_bounds__( NW * 64 ) gemmDecRows( const GemmArgs a )
		{
			// NW = waves that split K: 4, or 8 for the MLP down-projection (K = 4 d) of 33 .. 128 rows -- gemvFused's own split there (TUNE_GEMV_K8), same order
			constexpr int G = MT * CT;
			constexpr int GPW = ( G + NW - 1 ) / NW;
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) float redD[];	 // [NW][G * 4][64]

			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = tid >> 6;
			const int n0 = blockIdx.x * 16 * CT;
			const int m0 = blockIdx.y * 16 * MT;
			const int kPer = a.K / NW;
			const int kBeg = wave * kPer + ( lane >> 4 ) * 8;
			const int steps = kPer / 32;

			const f16* pw[ CT ];
	#pragma unroll
			for( int c = 0; c < CT; c++ )
			{
				int n = n0 + c * 16 + ( lane & 15 );
				n = n < a.N ? n : a.N - 1;
				pw[ c ] = a.W + (long long)n * a.K + kBeg;
			}
			const f16* px[ MT ];
	#pragma unroll
			for( int t = 0; t < MT; t++ )
			{
				int m = m0 + t * 16 + ( lane & 15 );
				m = m < a.M ? m : a.M - 1;
				px[ t ] = a.A + rowOffset( m, a.Mb, a.lda, a.aBatchStride ) + kBeg;
			}

			f32x4 acc[ MT ][ CT ];
	#pragma unroll
			for( int t = 0; t < MT; t++ )
	#pragma unroll
				for( int c = 0; c < CT; c++ ) acc[ t ][ c ] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };

			// software pipeline: the fragments of k-steps s + 1 .. s + DEPTH - 1 are in flight behind the MFMAs of step s (a ring of DEPTH register sets; the
			// smaller tiles have the registers for a deeper ring, and need it: fewer MFMAs per step to hide an L2 round trip behind)
			// (DEPTH: 2 at 64 x 64 -- 152 VGPRs already --, 3 at 64 x 32 / 32 x 64, 4 at 32 x 32; option dec_depth = 2 pins the round-5a pipeline for A/B runs)
			f16x8 fw[ DEPTH ][ CT ], fx[ DEPTH ][ MT ];
	#pragma unroll
			for( int d = 0; d < DEPTH - 1; d++ )
				if( d < steps )
				{
	#pragma unroll
					for( int c = 0; c < CT; c++ ) fw[ d ][ c ] = *(const f16x8*)( pw[ c ] + d * 32 );
	#pragma unroll
					for( int t = 0; t < MT; t++ ) fx[ d ][ t ] = *(const f16x8*)( px[ t ] + d * 32 );
				}
			for( int s0 = 0; s0 < steps; s0 += DEPTH )
			{
	#pragma unroll
				for( int u = 0; u < DEPTH; u++ )
				{
					const int s = s0 + u;
					if( s >= steps ) break;
					constexpr int ahead = DEPTH - 1;
					const int nxt = ( u + ahead ) % DEPTH;	  // the set step s - 1 has just released
					if( s + ahead < steps )
					{
	#pragma unroll
						for( int c = 0; c < CT; c++ ) fw[ nxt ][ c ] = *(const f16x8*)( pw[ c ] + ( s + ahead ) * 32 );
	#pragma unroll
						for( int t = 0; t < MT; t++ ) fx[ nxt ][ t ] = *(const f16x8*)( px[ t ] + ( s + ahead ) * 32 );
					}
	#pragma unroll
					for( int t = 0; t < MT; t++ )
	#pragma unroll
						for( int c = 0; c < CT; c++ )
							acc[ t ][ c ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( fw[ u ][ c ], fx[ u ][ t ], acc[ t ][ c ], 0, 0, 0 );
				}
			}

			// ---- the 4 K-quarters of the workgroup meet in LDS ----
	#pragma unroll
			for( int t = 0; t < MT; t++ )
	#pragma unroll
				for( int c = 0; c < CT; c++ )
	#pragma unroll
					for( int r = 0; r < 4; r++ ) redD[ ( wave * G * 4 + ( t * CT + c ) * 4 + r ) * 64 + lane ] = acc[ t ][ c ][ r ];
			__syncthreads();
			f32x4 part[ GPW ];
	#pragma unroll
			for( int i = 0; i < GPW; i++ )
			{
				const int g = wave + NW * i;
				if( g >= G ) continue;
	#pragma unroll
				for( int r = 0; r < 4; r++ )
				{
					float v = redD[ ( 0 * G * 4 + g * 4 + r ) * 64 + lane ];
	#pragma unroll
					for( int w = 1; w < NW; w++ ) v += redD[ ( w * G * 4 + g * 4 + r ) * 64 + lane ];
					part[ i ][ r ] = v;
				}
			}

			decRowsEpilogue<EPI, MT, CT, NW>( a, part, m0, n0, wave, lane );
		}

		// gemmDecTile (round 6): the same products (129 .. 512 rows) with the operands staged through LDS in FULL 128-byte lines. gemmDecRows' waves read their
		// fragments straight from L2, 16 rows x 64 bytes per instruction -- every 128-byte line is requested twice, by different instructions, and the kernel is bound by
		// that request stream (448 x 4096 x 1024: 21 us whatever the prefetch depth, 13 us at half the rows; profiles/r06_evidence/decode_rows_r6p.txt). Here a
		// workgroup (4 waves, 64 activation rows x 16 CT weight rows) walks K in tiles of 64 = one line per row: LDS-DMA pieces of 8 rows x 128 bytes (source chunk
		// XOR-swizzled as in the encoder's kernels), a ring of DT_NBUF tiles with counted waits and one barrier per tile, fragments by ds_read_b128. Wave w owns the
		// 16-column tile w % CT of MT / (4 / CT) row tiles -- the groups g = w + 4 i of decRowsEpilogue -- so its W fragment is the srcA operand of consecutive MFMAs.
		// THE SUMS ARE gemvFused's / gemmDecRows': those kernels give every wave a quarter of K and add the four partial tiles in wave order; here every wave walks all of
		// K, but closes an accumulator at each quarter of K and adds the four in the same order: ((P0 + P1) + P2) + P3, each Pi the same chain of k-steps of 32.
		constexpr int DT_NBUF = 4;
		__device__ __forceinline__ void ldsDmaOne( const void* base, unsigned off0, unsigned dst )
		{
			unsigned keep;
			asm volatile( "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
						  : "=&s"( keep )
						  : "s"( base ), "v"( off0 ), "s"( dst )
						  : "memory" );
		}
		// SPLIT = 8 (round 6, the MLP down-projection of 33 .. 128 rows: N = d gives only N / 32 = 32 .. 40 column tiles): blockIdx.y selects an EIGHTH of K instead of
		// a row tile -- gemvFused's eight-wave K split (TUNE_GEMV_K8) dealt to eight workgroups. The FP32 partial tile P_e goes to a.splitScratch[e][M][N]; the launch
		// that follows (decSplitCombine) adds the eight in gemvFused's order, ((P0 + P1) + ... ) + P7, then bias and residual: the same bits. Two launches, no atomics
		// (a last-arrival combine needs agent-scope fences: an L2 write-back per workgroup on gfx950, 5-30 us -- see gemmAllRows).
		template<int EPI, int CT, int KS = 1, int NBUF = DT_NBUF, int MT = 4, int SPLIT = 0>
		__global__ void __launch_bounds__( 256 ) gemmDecTile( const GemmArgs a )
		{
			// KS = K tiles of 64 per ring slot and barrier (2 for the deep products: K = 4096 is 64 tiles, and a tile is only 4 .. 8 MFMAs per wave)
			// MT = row tiles of 16 per workgroup: 4, or 6 / 8 with CT = 2 for the wide products of 65 .. 128 rows (all rows in one workgroup per 32 columns)
			static_assert( CT == 4 || CT == 2, "wave w owns column tile w % CT" );
			static_assert( MT == 4 || ( CT == 2 && ( MT == 6 || MT == 8 ) ), "an even number of row tiles per column pair" );
			constexpr int NW = 4, GPW = MT * CT / NW, AP = MT / 2;	  // AP = A pieces (8 rows x 128 bytes) per wave
			constexpr int A_BYTES = MT * 16 * 128, W_BYTES = CT * 16 * 128, TILE = A_BYTES + W_BYTES, STAGE = KS * TILE;
			constexpr int P = KS * ( AP + ( CT == 4 ? 2 : 1 ) );	  // load instructions per slot and wave
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemD[];
			typedef __attribute__( ( address_space( 3 ) ) ) void* LdsPtr;
			const int tid = threadIdx.x;
			const int lane = tid & 63;
			const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
			const int n0 = blockIdx.x * 16 * CT;
			const int m0 = SPLIT ? 0 : blockIdx.y * 16 * MT;
			const int kOff = SPLIT ? blockIdx.y * ( a.K / ( SPLIT ? SPLIT : 1 ) ) : 0;	 // first K element of this workgroup's share
			const int nk = SPLIT ? a.K / ( SPLIT ? SPLIT : 1 ) / 64 : a.K / 64, perQ = SPLIT ? nk : nk / 4, nSlots = nk / KS;
			const f16* const Ak = a.A + kOff;
			const f16* const Wk = a.W + kOff;

			// ---- producer: A = 2 MT pieces of 8 rows (wave w: pieces AP w .. AP w + AP - 1), W = 2 CT pieces (CT = 4: 2 w, 2 w + 1; CT = 2: piece w)
			const int rIn = lane >> 3, cPhys = lane & 7;
			unsigned offA[ AP ], offW[ 2 ];
	#pragma unroll
			for( int i = 0; i < AP; i++ )
			{
				const int row = ( wave * AP + i ) * 8 + rIn;
				const int cl = cPhys ^ ( ( row >> 1 ) & 7 );
				int m = m0 + row;
				m = m < a.M ? m : a.M - 1;
				offA[ i ] = (unsigned)( ( rowOffset( m, a.Mb, a.lda, a.aBatchStride ) + cl * 8 ) * 2 );
			}
	#pragma unroll
			for( int i = 0; i < 2; i++ )
			{
				const int rowW = CT == 4 ? ( wave * 2 + i ) * 8 + rIn : wave * 8 + rIn;
				const int clW = cPhys ^ ( ( rowW >> 1 ) & 7 );
				int n = n0 + rowW;
				n = n < a.N ? n : a.N - 1;
				offW[ i ] = (unsigned)( ( (long long)n * a.K + clW * 8 ) * 2 );
			}
			const unsigned ldsBase = __builtin_amdgcn_readfirstlane( (unsigned)(size_t)(LdsPtr)smemD );
			auto issue = [ & ]( int slot )
			{
	#pragma unroll
				for( int u = 0; u < KS; u++ )
				{
					const int kt = slot * KS + u;
					const unsigned buf = ldsBase + (unsigned)( slot % NBUF ) * STAGE + u * TILE;
					ldsDmaPair( Ak + kt * 64, offA[ 0 ], offA[ 1 ], buf + (unsigned)wave * ( AP * 1024u ) );
					if constexpr( AP == 3 ) ldsDmaOne( Ak + kt * 64, offA[ 2 ], buf + (unsigned)wave * ( AP * 1024u ) + 2048u );
					if constexpr( AP == 4 ) ldsDmaPair( Ak + kt * 64, offA[ 2 ], offA[ 3 ], buf + (unsigned)wave * ( AP * 1024u ) + 2048u );
					if constexpr( CT == 4 )
						ldsDmaPair( Wk + kt * 64, offW[ 0 ], offW[ 1 ], buf + A_BYTES + (unsigned)wave * 2048u );
					else
						ldsDmaOne( Wk + kt * 64, offW[ 0 ], buf + A_BYTES + (unsigned)wave * 1024u );
				}
			};

			// ---- consumer: lane l reads row l & 15 of a 16-row tile, logical chunk 4 h + (l >> 4), stored at chunk ^ ((row >> 1) & 7)
			const int cTile = wave % CT, tFirst = wave / CT;	 // group g = wave + 4 i: column tile g % CT = cTile, row tile g / CT = tFirst + ( 4 / CT ) i (G = MT CT is a multiple of 4)
			unsigned fragOff[ 2 ];
	#pragma unroll
			for( int h = 0; h < 2; h++ ) fragOff[ h ] = (unsigned)( ( lane & 15 ) * 128 + ( ( ( ( h << 2 ) + ( lane >> 4 ) ) ^ ( ( lane >> 1 ) & 7 ) ) << 4 ) );

			f32x4 acc[ GPW ], tot[ GPW ];
	#pragma unroll
			for( int i = 0; i < GPW; i++ ) acc[ i ] = tot[ i ] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };

	#pragma unroll
			for( int d = 0; d < NBUF - 1; d++ )
				if( d < nSlots ) issue( d );
			int inQ = 0, quarter = 0;
			for( int slot = 0; slot < nSlots; slot++ )
			{
				// slot `slot` has landed when no more than the pieces of the (up to NBUF - 2) younger slots are outstanding
				const int younger = min( NBUF - 2, nSlots - 1 - slot );
				static_assert( NBUF == 3 || NBUF == 4, "one or two younger slots" );
				if( younger >= 2 )
					asm volatile( "s_waitcnt vmcnt(%0)" ::"n"( 2 * P ) : "memory" );
				else if( younger == 1 )
					asm volatile( "s_waitcnt vmcnt(%0)" ::"n"( P ) : "memory" );
				else
					asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
				asm volatile( "s_barrier" ::: "memory" );	 // the slot is complete for every wave; every wave has issued the MFMAs of the slot before, whose buffer the next issue overwrites
				if( slot + NBUF - 1 < nSlots ) issue( slot + NBUF - 1 );
	#pragma unroll
				for( int u = 0; u < KS; u++ )
				{
					const unsigned char* const buf = smemD + ( slot % NBUF ) * STAGE + u * TILE;
					f16x8 fw[ 2 ], fx[ GPW ][ 2 ];
	#pragma unroll
					for( int h = 0; h < 2; h++ )
					{
						fw[ h ] = *(const f16x8*)( buf + A_BYTES + cTile * 2048 + fragOff[ h ] );
	#pragma unroll
						for( int i = 0; i < GPW; i++ ) fx[ i ][ h ] = *(const f16x8*)( buf + ( tFirst + ( 4 / CT ) * i ) * 2048 + fragOff[ h ] );
					}
	#pragma unroll
					for( int h = 0; h < 2; h++ )
	#pragma unroll
						for( int i = 0; i < GPW; i++ ) acc[ i ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( fw[ h ], fx[ i ][ h ], acc[ i ], 0, 0, 0 );
					if( ++inQ == perQ )
					{
						// a quarter of K is complete: the partial tile of gemvFused's wave `quarter`
	#pragma unroll
						for( int i = 0; i < GPW; i++ )
						{
	#pragma unroll
							for( int r = 0; r < 4; r++ ) tot[ i ][ r ] = quarter == 0 ? acc[ i ][ r ] : tot[ i ][ r ] + acc[ i ][ r ];
							acc[ i ] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
						}
						inQ = 0;
						quarter++;
					}
				}
			}
			if constexpr( SPLIT != 0 )
			{
				// the partial tile of this eighth of K: group g = wave + 4 i, lane = (activation row, four consecutive columns) as in decRowsEpilogue
				float* const part = a.splitScratch + (long long)blockIdx.y * a.M * a.N;
	#pragma unroll
				for( int i = 0; i < GPW; i++ )
				{
					const int g = wave + NW * i;
					const int t = g / CT, c = g - t * CT;
					const int mm = t * 16 + ( lane & 15 );
					const int nn = n0 + c * 16 + ( lane >> 4 ) * 4;
					if( mm < a.M && nn < a.N ) *(f32x4*)( part + (long long)mm * a.N + nn ) = tot[ i ];
				}
			}
			else
				decRowsEpilogue<EPI, MT, CT, NW>( a, tot, m0, n0, wave, lane );
		}

		// out[m][n] = ( ( ( P0 + P1 ) + ... + P7 ) + bias ) + res: the eight partial tiles of gemmDecTile<.., SPLIT = 8> in gemvFused's wave order, then epilogueOne<EPI_F32>'s order
		template<int SPLIT>
		__global__ void __launch_bounds__( 256 ) decSplitCombine( const GemmArgs a )
		{
			const int n4 = a.N >> 2;
			const int idx = blockIdx.x * 256 + threadIdx.x;
			if( idx >= a.M * n4 ) return;
			const int mm = idx / n4, nn = ( idx - mm * n4 ) * 4;
			const long long stride = (long long)a.M * a.N;
			const float* const p = a.splitScratch + (long long)mm * a.N + nn;
			f32x4 v[ SPLIT ];
	#pragma unroll
			for( int e = 0; e < SPLIT; e++ ) v[ e ] = *(const f32x4*)( p + e * stride );
			const long long off = (long long)mm * a.ldc + nn;
			f32x4 bv = { 0.0f, 0.0f, 0.0f, 0.0f }, rv = { 0.0f, 0.0f, 0.0f, 0.0f };
			if( a.bias ) bv = *(const f32x4*)( a.bias + nn );
			if( a.res ) rv = *(const f32x4*)( a.res + off );
			f32x4 o = v[ 0 ];
	#pragma unroll
			for( int e = 1; e < SPLIT; e++ )
	#pragma unroll
				for( int r = 0; r < 4; r++ ) o[ r ] += v[ e ][ r ];
			// (gemvFused adds its zero-initialised bias / residual registers when the pointers are null: so does this)
	#pragma unroll
			for( int r = 0; r < 4; r++ ) o[ r ] = ( o[ r ] + bv[ r ] ) + rv[ r ];
			*(f32x4*)( a.out32 + off ) = o;
		}
	}	// namespace

	template<int EPI, int MT>
	static int launchAllRowsT( const GemmArgs& a, int tiles, hipStream_t stream )
	{
		constexpr int lds = 4 * MT * 2 * 4 * 64 * 4;
		if( lds > 48 * 1024 )
		{
			static PerDeviceOnce once;
			if( const int onceDev = once.needed(); onceDev >= 0 )
			{
				WH_HIP( hipFuncSetAttribute( (const void*)gemmAllRows<EPI, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds ) );
				once.mark( onceDev );
			}
		}
		hipLaunchKernelGGL( ( gemmAllRows<EPI, MT> ), dim3( tiles ), dim3( 256 ), lds, stream, a );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	// 33 .. 128 rows, EPI_F32, A in global memory, at least 512 column tiles. Returns 1 when the shape is not covered.
	static int launchAllRows( const GemmArgs& a, hipStream_t stream )
	{
		const int tiles = ( a.N + 31 ) / 32;
		if( a.lnX || a.epi != EPI_F32 || a.M <= 32 || a.M > 128 || ( a.K % 128 ) != 0 || tiles < 512 ) return 1;
		switch( ( a.M + 15 ) / 16 )
		{
		case 3: return launchAllRowsT<EPI_F32, 3>( a, tiles, stream );
		case 4: return launchAllRowsT<EPI_F32, 4>( a, tiles, stream );
		case 5: return launchAllRowsT<EPI_F32, 5>( a, tiles, stream );
		case 6: return launchAllRowsT<EPI_F32, 6>( a, tiles, stream );
		case 7: return launchAllRowsT<EPI_F32, 7>( a, tiles, stream );
		default: return launchAllRowsT<EPI_F32, 8>( a, tiles, stream );
		}
	}

	template<int EPI, int MT, int CT, int DEPTH, int NW = 4>
	static int launchDecRowsD( const GemmArgs& a, hipStream_t stream )
	{
		constexpr int lds = NW * MT * CT * 4 * 64 * 4;
		if( lds > 48 * 1024 )
		{
			static PerDeviceOnce once;
			if( const int onceDev = once.needed(); onceDev >= 0 )
			{
				WH_HIP( hipFuncSetAttribute( (const void*)gemmDecRows<EPI, MT, CT, DEPTH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds ) );
				once.mark( onceDev );
			}
		}
		hipLaunchKernelGGL( ( gemmDecRows<EPI, MT, CT, DEPTH, NW> ), dim3( ( a.N + 16 * CT - 1 ) / ( 16 * CT ), ( a.M + 16 * MT - 1 ) / ( 16 * MT ) ), dim3( NW * 64 ), lds, stream, a );
		WH_HIP( hipGetLastError() );
		return 0;
	}
	template<int EPI, int MT, int CT>
	static int launchDecRowsK( const GemmArgs& a, hipStream_t stream )
	{
		constexpr int deep = MT * CT >= 16 ? 2 : ( MT * CT >= 8 ? 3 : 4 );
		if constexpr( deep != 2 )
			if( g_opt.decDepth != 2 ) return launchDecRowsD<EPI, MT, CT, deep>( a, stream );
		return launchDecRowsD<EPI, MT, CT, 2>( a, stream );
	}

	// gemmDecTile: K must divide into four quarters of whole 64-element tiles (the K split the sums follow); operands addressed as a 64-bit base + 32-bit offsets
	template<int EPI, int CT, int KS = 1, int NBUF = DT_NBUF, int MT = 4>
	static int launchDecTileK( const GemmArgs& a, hipStream_t stream )
	{
		constexpr int lds = NBUF * KS * ( MT * 16 * 128 + CT * 16 * 128 );
		if( lds > 48 * 1024 )
		{
			static PerDeviceOnce once;
			if( const int onceDev = once.needed(); onceDev >= 0 )
			{
				WH_HIP( hipFuncSetAttribute( (const void*)gemmDecTile<EPI, CT, KS, NBUF, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds ) );
				once.mark( onceDev );
			}
		}
		hipLaunchKernelGGL( ( gemmDecTile<EPI, CT, KS, NBUF, MT> ), dim3( ( a.N + 16 * CT - 1 ) / ( 16 * CT ), ( a.M + 16 * MT - 1 ) / ( 16 * MT ) ), dim3( 256 ), lds, stream, a );
		WH_HIP( hipGetLastError() );
		return 0;
	}
	static bool decTileOk( const GemmArgs& a )
	{
		const long long aBytes = 2ll * ( a.Mb > 0 && a.Mb < a.M ? ( (long long)( a.M / a.Mb ) + 1 ) * a.aBatchStride + (long long)a.Mb * a.lda : (long long)a.M * a.lda ) + 2ll * a.K;
		return ( a.K % 256 ) == 0 && ( a.lda % 8 ) == 0 && ( a.aBatchStride % 8 ) == 0 && aBytes < ( 1ll << 31 ) && 2ll * a.N * a.K < ( 1ll << 31 );
	}

	// Tile of a big-batch decode product: 64 x 64 (rows x columns) while that leaves enough workgroups for the chip, else 64 x 32, else 32 x 32.
	// Option dec_tile = <MT><CT> (44, 42, 24, 22) pins one for A/B runs and tests.
	template<int EPI>
	static int launchDecRowsT( const GemmArgs& a, hipStream_t stream )
	{
		const int pinned = g_opt.decTile;
		auto wgs = [ & ]( int mt, int ct ) { return ( ( a.N + 16 * ct - 1 ) / ( 16 * ct ) ) * ( ( a.M + 16 * mt - 1 ) / ( 16 * mt ) ); };
		// option dec_lds: the LDS-staged kernel (64 x 64, or 64 x 32 while the wider tile leaves fewer than 192 workgroups)
		// (measured, tools/gemv_time.py: 448 x 4096 x 1024 14.4 against 22.1 us, 448 x 1024 x 4096 21.2 / 26.7, 448 x 1024 x 1024 7.7 / 9.2, 224 x 4096 x 1024 10.8 / 14.1;
		// at 224 rows the N = 1024 products would get 128 workgroups of 64 x 32 and lose to gemmDecRows' 32 x 32 tiles: 7.6 / 6.6 and 20.5 / 18.1 us -- those keep it)
		if( g_opt.decLds == 1 && pinned == 0 && decTileOk( a ) )
		{
			if( wgs( 4, 4 ) >= 192 ) return launchDecTileK<EPI, 4>( a, stream );
			// (64 x 32 tiles from 160 workgroups: 320 x 1024 x 4096 17.5 against 30.9 us, 320 x 1024 x 1024 7.6 / 10.1; at 128 workgroups -- 224 / 256 rows -- 7.6 against 6.5 us)
			if( wgs( 4, 2 ) >= 160 )
			{
				// deep products (the MLP down-projection): two K tiles per ring slot and barrier when a quarter of K is an even number of tiles (448 x 1024 x 4096: 18.1 against 21.7 us)
				if( g_opt.decLdsKs == 2 && a.K >= 2048 && ( a.K % 512 ) == 0 ) return launchDecTileK<EPI, 2, 2, 3>( a, stream );
				return launchDecTileK<EPI, 2>( a, stream );
			}
		}
		int tile = pinned;
		if( tile != 44 && tile != 42 && tile != 24 && tile != 22 )
			tile = wgs( 4, 4 ) >= 192 ? 44 : ( wgs( 4, 2 ) >= 192 ? 42 : 22 );
		switch( tile )
		{
		case 44: return launchDecRowsK<EPI, 4, 4>( a, stream );
		case 42: return launchDecRowsK<EPI, 4, 2>( a, stream );
		case 24: return launchDecRowsK<EPI, 2, 4>( a, stream );
		default: return launchDecRowsK<EPI, 2, 2>( a, stream );
		}
	}

	// 33 .. 128 rows against a WIDE weight matrix (N >= 2048: the MLP up-projection, the fused QKV product): ALL rows in one row tile of 16 MT rows and 32
	// columns per workgroup. gemvFused's 16-column workgroups re-read the rows once per 16 columns: at 70 rows and N = 4096 that is 82 MB of L2 -> CU
	// traffic for 8 MB of weights (15 us per launch); here 29 MB over N / 32 workgroups. Same K split and summation order: the same bits.
	template<int EPI>
	static int launchDecRowsOneTile( const GemmArgs& a, hipStream_t stream )
	{
		// option dec_lds (round 6): the LDS-staged kernel with all rows in one workgroup per 32 columns (4 / 6 / 8 row tiles), the same sums
		if( g_opt.decLds == 1 && g_opt.decTile == 0 && decTileOk( a ) )
		{
			const int mt = ( a.M + 15 ) / 16;
			// two K tiles per ring slot and barrier (dec_lds_ks 2, the default) for 4 and 6 row tiles: 40 x 5120 x 1280 8.9 -> 7.8 us, 70 rows 10.9 -> 10.0; level at 8
			// row tiles (123 KiB of LDS), and SLOWER for the K-split instances (K = 5120 at 70 rows: 11.9 -> 13.9) and the vocabulary product (38 -> 48 us: one
			// workgroup per CU instead of three) -- those keep one tile per slot (profiles/r06_evidence/small_batch_products.txt)
			if( g_opt.decLdsKs == 2 && ( a.K % 128 ) == 0 && mt <= 6 )
				return mt <= 4 ? launchDecTileK<EPI, 2, 2, 3, 4>( a, stream ) : launchDecTileK<EPI, 2, 2, 3, 6>( a, stream );
			if( mt <= 4 ) return launchDecTileK<EPI, 2, 1, DT_NBUF, 4>( a, stream );
			if( mt <= 6 ) return launchDecTileK<EPI, 2, 1, DT_NBUF, 6>( a, stream );
			return launchDecTileK<EPI, 2, 1, DT_NBUF, 8>( a, stream );
		}
		switch( ( a.M + 15 ) / 16 )
		{
		case 3: return launchDecRowsK<EPI, 3, 2>( a, stream );
		case 4: return launchDecRowsK<EPI, 4, 2>( a, stream );
		case 5: return launchDecRowsK<EPI, 5, 2>( a, stream );
		case 6: return launchDecRowsK<EPI, 6, 2>( a, stream );
		case 7: return launchDecRowsK<EPI, 7, 2>( a, stream );
		default: return launchDecRowsK<EPI, 8, 2>( a, stream );
		}
	}
	// 33 .. 128 rows against a NARROW, DEEP weight matrix (N <= 2048, K >= 2048: the MLP down-projection): 16 columns x all rows per workgroup, EIGHT waves
	// splitting K -- gemvFused's own split for this product (TUNE_GEMV_K8), so the same bits -- instead of its 16 columns x 32 rows with the rows re-read per group
	static int launchDecRowsDeep( const GemmArgs& a, hipStream_t stream )
	{
		if( a.lnX || a.epi != EPI_F32 || a.M <= 32 || a.M > GEMV_FUSED_MAX_ROWS || a.N > 2048 || ( a.N % 16 ) != 0 || a.K < 2048 || ( a.K % 256 ) != 0 || !( g_tuning & TUNE_GEMV_K8 ) ) return 1;
		switch( ( a.M + 15 ) / 16 )
		{
		case 3: return launchDecRowsD<EPI_F32, 3, 1, 4, 8>( a, stream );
		case 4: return launchDecRowsD<EPI_F32, 4, 1, 4, 8>( a, stream );
		case 5: return launchDecRowsD<EPI_F32, 5, 1, 4, 8>( a, stream );
		case 6: return launchDecRowsD<EPI_F32, 6, 1, 4, 8>( a, stream );
		case 7: return launchDecRowsD<EPI_F32, 7, 1, 4, 8>( a, stream );
		default: return launchDecRowsD<EPI_F32, 8, 1, 3, 8>( a, stream );
		}
	}
	// 33 .. 128 rows against a NARROW, DEEP weight matrix, option dec_split (round 6): the eight K shares of gemvFused's eight waves dealt to eight workgroups of the
	// LDS-staged kernel per 32 columns (N / 32 x 8 = 256 .. 320 workgroups instead of gemvFused's N / 16 x 2 re-reading the rows per 16 columns), the eight partial tiles
	// added by a second launch in wave order: the same bits. Needs the context's scratch (8 x M x N floats). Returns 1 when the shape is not covered.
	static bool decSplitShape( const GemmArgs& a )
	{
		// (measured and not kept: LayerNorm of the finished rows for the next product inside the combine launch, a wave per row -- 10 workgroups at 40 rows take 6.2 us
		// against 2.5 + 5.3 for the two launches it replaces, the beam job did not move: 1271 against 1270 audio-s/s)
		return !( a.lnX || a.epi != EPI_F32 || !a.splitScratch || a.M <= 32 || a.M > GEMV_FUSED_MAX_ROWS || a.N > 2048 || ( a.N % 32 ) != 0 || a.K < 2048 || ( a.K % 512 ) != 0 || a.Mb < a.M ||
			!( g_tuning & TUNE_GEMV_K8 ) || !decTileOk( a ) );
	}
	static int launchDecRowsSplit( const GemmArgs& a, hipStream_t stream )
	{
		if( !decSplitShape( a ) ) return 1;
		auto go = [ & ]( auto mtTag ) -> int
		{
			constexpr int MT = decltype( mtTag )::value;
			constexpr int lds = DT_NBUF * ( MT * 16 * 128 + 2 * 16 * 128 );
			if( lds > 48 * 1024 )
			{
				static PerDeviceOnce once;
				if( const int onceDev = once.needed(); onceDev >= 0 )
				{
					WH_HIP( hipFuncSetAttribute( (const void*)gemmDecTile<EPI_F32, 2, 1, DT_NBUF, MT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds ) );
					once.mark( onceDev );
				}
			}
			hipLaunchKernelGGL( ( gemmDecTile<EPI_F32, 2, 1, DT_NBUF, MT, 8> ), dim3( a.N / 32, 8 ), dim3( 256 ), lds, stream, a );
			WH_HIP( hipGetLastError() );
			hipLaunchKernelGGL( ( decSplitCombine<8> ), dim3( ( a.M * ( a.N / 4 ) + 255 ) / 256 ), dim3( 256 ), 0, stream, a );
			WH_HIP( hipGetLastError() );
			return 0;
		};
		const int mt = ( a.M + 15 ) / 16;
		if( mt <= 4 ) return go( std::integral_constant<int, 4>{} );
		if( mt <= 6 ) return go( std::integral_constant<int, 6>{} );
		return go( std::integral_constant<int, 8>{} );
	}
	// returns 1 when the shape is not one of these
	static int launchDecRowsWide( const GemmArgs& a, hipStream_t stream )
	{
		if( a.lnX || a.M <= 32 || a.M > GEMV_FUSED_MAX_ROWS || a.N < 2048 || ( a.K % 128 ) != 0 || a.K > 2048 ) return 1;
		switch( a.epi )
		{
		case EPI_F16_GELU: return launchDecRowsOneTile<EPI_F16_GELU>( a, stream );
		case EPI_QKV_DEC: return launchDecRowsOneTile<EPI_QKV_DEC>( a, stream );
		case EPI_F32: if( g_opt.decWideRows == 2 ) return launchDecRowsOneTile<EPI_F32>( a, stream ); break;	 // (diagnostic: the accumulators of the one-tile instances in FP32)
		}
		return 1;
	}

	// 129 .. GEMV_MAX_ROWS rows, A in global memory (a LayerNorm in front is its own launch at this many rows)
	static int launchDecRows( const GemmArgs& a, hipStream_t stream )
	{
		if( a.lnX || ( a.K % 128 ) != 0 )
		{
			setError( "gemv: more than 128 rows need FP16 activation rows and K a multiple of 128" );
			return -1;
		}
		switch( a.epi )
		{
		case EPI_F32: return launchDecRowsT<EPI_F32>( a, stream );
		case EPI_F16_GELU: return launchDecRowsT<EPI_F16_GELU>( a, stream );
		case EPI_QKV_DEC: return launchDecRowsT<EPI_QKV_DEC>( a, stream );
		case EPI_Q_DEC: return launchDecRowsT<EPI_Q_DEC>( a, stream );
		}
		setError( "gemv: epilogue not available" );
		return -1;
	}

	template<int EPI, int PRO, int ROWS, int NW, int UNROLL, int MT>
	static int launchGemvK( const GemmArgs& a, hipStream_t stream )
	{
		const size_t lds = PRO != 0 ? (size_t)16 * MT * GV_XS_STRIDE * sizeof( f16 ) : 0;
		if( lds > 64 * 1024 )
		{
			static PerDeviceOnce once;
			if( const int onceDev = once.needed(); onceDev >= 0 )
			{
				WH_HIP( hipFuncSetAttribute( (const void*)gemvFused<EPI, PRO, ROWS, NW, UNROLL, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds ) );
				once.mark( onceDev );
			}
		}
		const int groups = ( a.M + 16 * MT - 1 ) / ( 16 * MT );
		hipLaunchKernelGGL( ( gemvFused<EPI, PRO, ROWS, NW, UNROLL, MT> ), dim3( ( a.N + ROWS - 1 ) / ROWS, groups ), dim3( NW * 64 ), lds, stream, a );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	template<int EPI, int PRO, int ROWS = 16, int NW = 4>
	static int launchGemvT( const GemmArgs& a, hipStream_t stream )
	{
		// a wave holds K / NW / 32 weight fragments; when they fit in 8 slots the 8-slot instance does the same work with
		// half the registers, which lets kernels of concurrent decode chains share a CU. More than 16 activation rows
		// (up to 32) take a second MFMA column tile per weight fragment.
		const bool small = a.K / NW / 32 <= 8 && ( g_tuning & TUNE_GEMV_SMALLREG );
		if constexpr( PRO == 0 )
		{
			// 33 .. 128 rows: four MFMA column tiles per weight fragment (64 rows per workgroup, two row groups beyond that);
			// always the 8-slot instance -- 4 x 8 activation fragments in flight are 128 registers
			if( a.M > 32 )
			{
				// 64 rows per workgroup read each weight row once per 64 rows, but N / ROWS x ceil(M / 64) workgroups must still
				// cover the chip: below 256 of them, 32 rows per workgroup (twice the workgroups, each with half the activation
				// traffic) measured 5.8 vs 7.8 us (N = K = 1024) and 14.1 vs 22.0 us (N = 1024, K = 4096) at 112 rows
				// TUNE_GEMV_MT8 (A/B): ALL rows in one workgroup when N / ROWS alone fills the chip (the MLP up-projection, N = 4096): the weights are
				// streamed once instead of once per 64 rows; 4 fragment slots instead of 8 keep 8 x 4 activation fragments at 128 registers
				if constexpr( NW == 4 )
					if( a.M > 64 && ( a.N + ROWS - 1 ) / ROWS >= 256 && ( g_tuning & TUNE_GEMV_MT8 ) ) return launchGemvK<EPI, PRO, ROWS, NW, 4, 8>( a, stream );
				const int wgs = ( a.N + ROWS - 1 ) / ROWS * ( ( a.M + 63 ) / 64 );
				if( wgs < 256 && ( g_tuning & TUNE_GEMV_ROWGROUPS ) ) return launchGemvK<EPI, PRO, ROWS, NW, 8, 2>( a, stream );
				return launchGemvK<EPI, PRO, ROWS, NW, 8, 4>( a, stream );
			}
		}
		if( a.M > 16 )
			return small ? launchGemvK<EPI, PRO, ROWS, NW, 8, 2>( a, stream ) : launchGemvK<EPI, PRO, ROWS, NW, GV_UNROLL_MAX, 2>( a, stream );
		return small ? launchGemvK<EPI, PRO, ROWS, NW, 8, 1>( a, stream ) : launchGemvK<EPI, PRO, ROWS, NW, GV_UNROLL_MAX, 1>( a, stream );
	}

	int launchGemv( const GemmArgs& a, hipStream_t stream )
	{
		if( a.M <= 0 || a.M > GEMV_MAX_ROWS || a.N <= 0 || a.K <= 0 || ( a.K % 128 ) != 0 )
		{
			setError( "gemv: need 0 < M <= 512 and K a multiple of 128" );
			return -1;
		}
		// more than 128 rows (a lock-step batch of 129 .. 512 sequences): 64 x 64 output tiles per workgroup (gemmDecRows).
		// Option dec_tile = 1 keeps gemvFused (16 columns x 64 rows per workgroup, row groups in blockIdx.y) for A/B runs.
		if( a.M > GEMV_FUSED_MAX_ROWS && ( g_opt.decTile != 1 || a.lnX ) ) return launchDecRows( a, stream );
		const bool ln = a.lnX != nullptr;
		// option dec_wide_rows: 33 .. 128 rows against N >= 2048 in one row tile per 32 columns (gemmDecRows) instead of gemvFused's 16-column workgroups
		if( a.M > 32 && !ln && g_opt.decWideRows )
		{
			const int rc = launchDecRowsWide( a, stream );
			if( rc <= 0 ) return rc;
		}
		// option dec_split: the same product with the eight K shares on eight workgroups of gemmDecTile and a combine launch
		if( a.M > 32 && !ln && g_opt.decSplit )
		{
			const int rc = launchDecRowsSplit( a, stream );
			if( rc <= 0 ) return rc;
		}
		// option dec_deep_rows: 33 .. 128 rows against N <= 2048, K >= 2048 (MLP down-projection) with all rows per 16-column workgroup and 8 waves over K
		if( a.M > 32 && !ln && g_opt.decDeepRows )
		{
			const int rc = launchDecRowsDeep( a, stream );
			if( rc <= 0 ) return rc;
		}
		// option vocab_lds (round 6): the vocabulary product (N / 32 >= 512) of 33 .. 128 rows as 64 x 64 tiles of the LDS-staged kernel (one or two row tiles): the rows
		// are re-read once per 64 columns instead of gemmAllRows' once per 32, the second row tile finds the weights in the Infinity Cache; the same K quarters added in
		// the same order (40 x 51865 x 1280: 37.7 against 67.1 us, 128 rows: 67.7 against 132.8)
		if( a.M > 32 && a.M <= GEMV_FUSED_MAX_ROWS && !ln && a.epi == EPI_F32 && ( a.N + 31 ) / 32 >= 512 && a.Mb >= a.M && g_opt.vocabLds == 1 && ( g_tuning & TUNE_GEMV_ALLROWS ) && decTileOk( a ) )
			return launchDecTileK<EPI_F32, 4>( a, stream );
		if( a.M > 32 && !ln && ( g_tuning & TUNE_GEMV_ALLROWS ) )
		{
			const int rc = launchAllRows( a, stream );
			if( rc <= 0 ) return rc;
		}
		if( ln && ( a.K > GV_MAXK_LN || a.M > 32 ) )
		{
			setError( "gemv: the fused LayerNorm prologue supports up to 32 rows of up to 1280 columns" );
			return -1;
		}
		// small N, large K (the MLP down projection): 4 weight rows per workgroup so that every CU streams
		// (up to 16 activation rows: beyond that the rows a workgroup re-reads outweigh its 4 weight rows, measured +3 % without)
		const bool rows4 = !ln && a.epi == EPI_F32 && ( a.N % 16 ) == 0 && a.N <= 2048 && a.K >= 2048 && a.M <= 16 && ( g_tuning & TUNE_GEMV_ROWS4 );
		// K >= 2048 (the MLP down-projection, 64 workgroups): 8 waves split K, so a wave's 16 weight fragments are ONE round of loads
		const bool k8 = !ln && !rows4 && a.epi == EPI_F32 && a.K >= 2048 && ( a.K % 256 ) == 0 && ( g_tuning & TUNE_GEMV_K8 );
		// more than 16 rows: the LayerNorm prologue is done by the whole workgroup, 16 rows at a time
		const bool lnBlock = ln && a.M > 16;
		switch( a.epi )
		{
		case EPI_F32:
			if( lnBlock ) return launchGemvK<EPI_F32, 2, 16, 4, 8, 2>( a, stream );
			if( ln ) return launchGemvT<EPI_F32, 1>( a, stream );
			if( k8 ) return launchGemvT<EPI_F32, 0, 16, 8>( a, stream );
			return rows4 ? launchGemvT<EPI_F32, 0, 4, 4>( a, stream ) : launchGemvT<EPI_F32, 0>( a, stream );
		case EPI_F16_GELU:
			if( lnBlock ) return launchGemvK<EPI_F16_GELU, 2, 16, 4, 8, 2>( a, stream );
			return ln ? launchGemvT<EPI_F16_GELU, 1>( a, stream ) : launchGemvT<EPI_F16_GELU, 0>( a, stream );
		case EPI_QKV_DEC:
			if( lnBlock ) return launchGemvK<EPI_QKV_DEC, 2, 16, 4, 8, 2>( a, stream );
			return ln ? launchGemvT<EPI_QKV_DEC, 1>( a, stream ) : launchGemvT<EPI_QKV_DEC, 0>( a, stream );
		case EPI_Q_DEC:
			if( lnBlock ) return launchGemvK<EPI_Q_DEC, 2, 16, 4, 8, 2>( a, stream );
			return ln ? launchGemvT<EPI_Q_DEC, 1>( a, stream ) : launchGemvT<EPI_Q_DEC, 0>( a, stream );
		}
		setError( "gemv: epilogue not available" );
		return -1;
	}

	template<int EPI, class C, bool WIDE>
	static int launchTiledK( const GemmArgs& b, hipStream_t stream )
	{
		static PerDeviceOnce once;
		if( const int onceDev = once.needed(); onceDev >= 0 )
		{
			WH_HIP( hipFuncSetAttribute( (const void*)gemmTiled<EPI, C, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES ) );
			once.mark( onceDev );
		}
		const int tilesM = ( b.M + C::BM - 1 ) / C::BM, tilesN = ( b.N + C::BN - 1 ) / C::BN;
		hipLaunchKernelGGL( ( gemmTiled<EPI, C, WIDE> ), dim3( tilesM * tilesN ), dim3( C::NT ), C::LDS_BYTES, stream, b );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	template<int EPI, class C = CfgDefault>
	static int launchTiledT( const GemmArgs& a, hipStream_t stream )
	{
		GemmArgs b = a;
		if( b.groupM == 0 ) b.groupM = ( g_tuning & TUNE_GEMM_GROUP_M ) ? ( C::BM >= 256 ? 4 : 8 ) : 1;
		// the LDS-transposed epilogue with 16-byte stores needs whole, aligned chunks
		bool wide = false;
		constexpr bool canWide = C::GL && C::TI == 2 && C::TJ == 2 &&
			( EPI == EPI_F32 || EPI == EPI_F16_GELU || EPI == EPI_CONV2 || EPI == EPI_QKV_ENC || EPI == EPI_CROSS_KV );
		if( canWide && ( g_tuning & TUNE_GEMM_WIDE_EPI ) )
		{
			const bool al16 = ( a.N % 8 ) == 0 && ( a.ldc % 8 ) == 0 && ( a.cBatchStride % 8 ) == 0;
			switch( EPI )
			{
			case EPI_F32: wide = al16 && ( ( (size_t)a.out32 | (size_t)a.res ) % 16 ) == 0; break;
			case EPI_CONV2: wide = al16 && ( ( (size_t)a.out32 | (size_t)a.pe ) % 16 ) == 0; break;
			case EPI_F16_GELU: wide = al16 && ( (size_t)a.out16 % 16 ) == 0; break;
			case EPI_QKV_ENC: wide = ( a.N % 64 ) == 0 && ( ( (size_t)a.q | (size_t)a.k ) % 16 ) == 0; break;
			case EPI_CROSS_KV: wide = ( a.N % 64 ) == 0 && ( ( (size_t)a.k | (size_t)a.v ) % 16 ) == 0; break;
			default: break;
			}
		}
		b.wideEpi = wide ? 1 : 0;
		if constexpr( canWide )
		{
			if( wide ) return launchTiledK<EPI, C, true>( b, stream );
		}
		return launchTiledK<EPI, C, false>( b, stream );
	}

	// What epilogueFast4 relies on (interior tiles of the two persistent kernels): a wave's 128 rows cross at most one segment boundary, and
	// everything it adds per lane fits 32 bits
	template<int EPI>
	static bool fastEpilogueOk( const GemmArgs& a )
	{
		if( EPI == EPI_QKV_ENC || EPI == EPI_CROSS_KV )
			return a.T >= 128 && ( a.H * HEAD_DIM ) % 128 == 0 && (long long)( a.H - 1 ) * a.T * 128 < ( 1ll << 31 );
		if( EPI != EPI_F32 && EPI != EPI_F16_GELU ) return false;
		const int es = EPI == EPI_F32 ? 4 : 2;
		bool fast = (long long)a.ldc * es * 128 < ( 1ll << 31 );
		if( a.Mb > 0 && a.Mb < a.M )
		{
			const long long cross = ( a.cBatchStride - (long long)a.Mb * a.ldc ) * es;
			fast = fast && a.Mb >= 128 && cross >= 0 && cross + (long long)a.ldc * es * 128 < ( 1ll << 31 );
		}
		return fast;
	}

	template<int EPI, bool WIDE, bool MF16 = false>
	static int launchTiled8K( const GemmArgs& b, hipStream_t stream )
	{
		static PerDeviceOnce once;
		static int cusOfDevice[ 64 ];
		int dev = 0;
		if( hipGetDevice( &dev ) != hipSuccess ) dev = 0;
		if( const int onceDev = once.needed(); onceDev >= 0 )
		{
			WH_HIP( hipFuncSetAttribute( (const void*)gemmTiled8<EPI, WIDE, MF16>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg8::LDS_BYTES ) );
			int cus = 0;
			WH_HIP( hipDeviceGetAttribute( &cus, hipDeviceAttributeMultiprocessorCount, dev ) );
			cusOfDevice[ onceDev ] = cus;
			once.mark( onceDev );
		}
		// persistent: one workgroup per CU (a workgroup takes all 160 KiB of LDS), each walks its share of the tiles
		const int tilesM = ( b.M + Cfg8::BM - 1 ) / Cfg8::BM, tilesN = ( b.N + Cfg8::BN - 1 ) / Cfg8::BN;
		int cus = cusOfDevice[ dev & 63 ] > 0 ? cusOfDevice[ dev & 63 ] : 256;
		if( b.cuLimit > 0 && b.cuLimit < cus ) cus = b.cuLimit;
		const int grid = tilesM * tilesN < cus ? tilesM * tilesN : cus;
		hipLaunchKernelGGL( ( gemmTiled8<EPI, WIDE, MF16> ), dim3( grid ), dim3( Cfg8::NT ), Cfg8::LDS_BYTES, stream, b );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	// the 8-wave 256x256x64 kernel; same preconditions for the LDS-transposed epilogue as launchTiledT
	template<int EPI, bool MF16 = false>
	static int launchTiled8( const GemmArgs& a, hipStream_t stream )
	{
		GemmArgs b = a;
		// WH_GEMM_GROUP_M: M tiles per band of the walk, for A/B runs (4: the band's A rows are 2 MB of an XCD's 4 MB L2 at K = 1024 and W is re-streamed once per band)
		static const int groupEnv = []() { const char* e = getenv( "WH_GEMM_GROUP_M" ); const int v = e ? atoi( e ) : 0; return v >= 1 && v <= 64 ? v : 0; }();
		if( b.groupM == 0 ) b.groupM = groupEnv ? groupEnv : ( ( g_tuning & TUNE_GEMM_GROUP_M ) ? 4 : 1 );
		bool wide = false;
		if( g_tuning & TUNE_GEMM_WIDE_EPI )
		{
			const bool al16 = ( a.N % 8 ) == 0 && ( a.ldc % 8 ) == 0 && ( a.cBatchStride % 8 ) == 0;
			switch( EPI )
			{
			case EPI_F32: wide = al16 && ( ( (size_t)a.out32 | (size_t)a.res ) % 16 ) == 0; break;
			case EPI_CONV2: wide = al16 && ( ( (size_t)a.out32 | (size_t)a.pe ) % 16 ) == 0; break;
			case EPI_F16_GELU: wide = al16 && ( (size_t)a.out16 % 16 ) == 0; break;
			case EPI_QKV_ENC: wide = ( a.N % 64 ) == 0 && ( ( (size_t)a.q | (size_t)a.k ) % 16 ) == 0; break;
			case EPI_CROSS_KV: wide = ( a.N % 64 ) == 0 && ( ( (size_t)a.k | (size_t)a.v ) % 16 ) == 0; break;
			default: break;
			}
		}
		b.wideEpi = wide ? 1 : 0;
		if( wide && ( g_tuning & TUNE_GEMM_FAST_EPI ) && ( EPI != EPI_QKV_ENC || ( a.T % 4 ) == 0 ) && fastEpilogueOk<EPI>( a ) ) b.wideEpi = 2;
		return wide ? launchTiled8K<EPI, true, MF16>( b, stream ) : launchTiled8K<EPI, false, MF16>( b, stream );
	}

	template<int EPI, bool WIDE, int SCH = 0>
	static int launchTiled4K( const GemmArgs& b, hipStream_t stream )
	{
		static PerDeviceOnce once;
		static int cusOfDevice[ 64 ];
		int dev = 0;
		if( hipGetDevice( &dev ) != hipSuccess ) dev = 0;
		if( const int onceDev = once.needed(); onceDev >= 0 )
		{
			WH_HIP( hipFuncSetAttribute( (const void*)gemmTiled4<EPI, WIDE, SCH>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg4::LDS_BYTES ) );
			int cus = 0;
			WH_HIP( hipDeviceGetAttribute( &cus, hipDeviceAttributeMultiprocessorCount, dev ) );
			cusOfDevice[ onceDev ] = cus;
			once.mark( onceDev );
		}
		// persistent: one workgroup per CU (512 registers per lane: one wave per SIMD), each walks its share of the tiles
		const int tilesM = ( b.M + Cfg4::BM - 1 ) / Cfg4::BM, tilesN = ( b.N + Cfg4::BN - 1 ) / Cfg4::BN;
		int cus = cusOfDevice[ dev & 63 ] > 0 ? cusOfDevice[ dev & 63 ] : 256;
		if( b.cuLimit > 0 && b.cuLimit < cus ) cus = b.cuLimit;
		const int grid = tilesM * tilesN < cus ? tilesM * tilesN : cus;
		hipLaunchKernelGGL( ( gemmTiled4<EPI, WIDE, SCH> ), dim3( grid ), dim3( Cfg4::NT ), Cfg4::LDS_BYTES, stream, b );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	// the 4-wave 256x256x64 kernel; preconditions of the LDS-transposed epilogue as launchTiled8, plus T % 4 == 0 for the V columns of the encoder's Q/K/V product
	template<int EPI, int SCH = 0>
	static int launchTiled4( const GemmArgs& a, hipStream_t stream )
	{
		GemmArgs b = a;
		static const int groupEnv = []() { const char* e = getenv( "WH_GEMM_GROUP_M" ); const int v = e ? atoi( e ) : 0; return v >= 1 && v <= 64 ? v : 0; }();
		if( b.groupM == 0 ) b.groupM = groupEnv ? groupEnv : ( ( g_tuning & TUNE_GEMM_GROUP_M ) ? 4 : 1 );
		bool wide = false;
		if( g_tuning & TUNE_GEMM_WIDE_EPI )
		{
			const bool al16 = ( a.N % 8 ) == 0 && ( a.ldc % 8 ) == 0 && ( a.cBatchStride % 8 ) == 0;
			switch( EPI )
			{
			case EPI_F32: wide = al16 && ( ( (size_t)a.out32 | (size_t)a.res ) % 16 ) == 0; break;
			case EPI_CONV2: wide = al16 && ( ( (size_t)a.out32 | (size_t)a.pe ) % 16 ) == 0; break;
			case EPI_F16_GELU: wide = al16 && ( (size_t)a.out16 % 16 ) == 0; break;
			case EPI_QKV_ENC: wide = ( a.N % 64 ) == 0 && ( a.T % 4 ) == 0 && ( ( (size_t)a.q | (size_t)a.k | (size_t)a.v ) % 16 ) == 0; break;
			case EPI_CROSS_KV: wide = ( a.N % 64 ) == 0 && ( ( (size_t)a.k | (size_t)a.v ) % 16 ) == 0; break;
			default: break;
			}
		}
		b.wideEpi = wide ? 1 : 0;
		if( wide && fastEpilogueOk<EPI>( a ) ) b.wideEpi = 2;
		return wide ? launchTiled4K<EPI, true, SCH>( b, stream ) : launchTiled4K<EPI, false, SCH>( b, stream );
	}

	// Tile-shape experiments on the plain FP32 epilogue (tools/gemm_probe.py): variant -> configuration
	int launchGemmVariant( const GemmArgs& a, int variant, hipStream_t stream )
	{
		switch( variant )
		{
		case 40: return launchTiled8<EPI_F32>( a, stream );	   // the 8-wave persistent kernel (round 3)
		case 52: return launchTiled8<EPI_F32, true>( a, stream );	   // the same with v_mfma_f32_16x16x32_f16 in the K loop (round 6)
		case 50: return launchTiled4<EPI_F32>( a, stream );	   // the 4-wave persistent kernel (round 4)
		case 25: return launchTiledT<EPI_F32, TileCfg<256, 256, 64, 4, 1, true, 2, 2, 2, 1>>( a, stream );	   // the 16-wave kernel of round 2 (products below gemmTiled8's threshold)
		case 26: return launchTiledT<EPI_F32, TileCfg<128, 128, 32, 3, 1, true, 2, 2, 2, 1>>( a, stream );
		case 2: return launchTiledT<EPI_F32, TileCfg<128, 128, 32, 3, 1>>( a, stream );	   // register-staged 128x128x32: what wh_debug_probe checks every variant against
#ifdef WH_PROBES
		case 51: return launchTiled4<EPI_F32, 16384>( a, stream );	   // correct: without the early W pieces / the counted wait after the epilogue
		// Everything below exists for tools/*probe*: tile-shape experiments (all correct). The shipped objects do not contain them: build with
		// WH_PROBES=1 python -m whisper_amd.build --force to get them back. (The ABLATIONS of rounds 2-4 -- kernels with loads, fragment reads,
		// MFMAs or stores removed to see what the rest costs: profiles/r02_gemm_kloop_ablation.txt, r03_gemm8_ablation.txt, r04_gemm4_probe.txt --
		// lived in the production kernels' source as compile-time branches until round 5; they are in the history up to commit 7317048.)
		case 27: return launchTiledT<EPI_F32, TileCfg<256, 256, 32, 4, 1, true, 2, 2, 3, 1>>( a, stream );
		case 20: return launchTiledT<EPI_F32, TileCfg<256, 256, 32, 4, 1, true, 2, 2, 3>>( a, stream );
		case 21: return launchTiledT<EPI_F32, TileCfg<256, 256, 32, 4, 1, true, 2, 2, 4>>( a, stream );
		case 22: return launchTiledT<EPI_F32, TileCfg<256, 128, 64, 2, 1, true, 2, 2, 3>>( a, stream );
		case 23: return launchTiledT<EPI_F32, TileCfg<256, 256, 32, 4, 1, true, 2, 2, 2>>( a, stream );
		case 24: return launchTiledT<EPI_F32, TileCfg<128, 256, 64, 2, 1, true, 2, 2, 3>>( a, stream );
		case 10: return launchTiledT<EPI_F32, TileCfg<128, 128, 64, 2, 1, true>>( a, stream );
		case 11: return launchTiledT<EPI_F32, TileCfg<128, 128, 32, 3, 1, true>>( a, stream );
		case 12: return launchTiledT<EPI_F32, TileCfg<256, 256, 64, 4, 1, true>>( a, stream );
		case 13: return launchTiledT<EPI_F32, TileCfg<256, 128, 64, 2, 1, true>>( a, stream );
		case 14: return launchTiledT<EPI_F32, TileCfg<256, 128, 32, 2, 1, true, 4, 2>>( a, stream );
		case 15: return launchTiledT<EPI_F32, TileCfg<256, 128, 64, 1, 1, true, 4, 2>>( a, stream );
		case 16: return launchTiledT<EPI_F32, TileCfg<256, 256, 64, 2, 1, true, 4, 2>>( a, stream );
		case 17: return launchTiledT<EPI_F32, TileCfg<256, 256, 32, 2, 1, true, 4, 2>>( a, stream );
		case 18: return launchTiledT<EPI_F32, TileCfg<128, 256, 32, 2, 1, true, 2, 4>>( a, stream );
		case 0: return launchTiledT<EPI_F32, TileCfg<128, 128, 64, 2, 2>>( a, stream );
		case 9: return launchTiledT<EPI_F32, TileCfg<128, 128, 32, 3, 1>>( a, stream );
		case 1: return launchTiledT<EPI_F32, TileCfg<128, 128, 64, 2, 1>>( a, stream );
		case 3: return launchTiledT<EPI_F32, TileCfg<256, 128, 64, 2, 1>>( a, stream );
		case 4: return launchTiledT<EPI_F32, TileCfg<256, 128, 64, 2, 2>>( a, stream );
		case 5: return launchTiledT<EPI_F32, TileCfg<256, 128, 32, 4, 1>>( a, stream );
		case 6: return launchTiledT<EPI_F32, TileCfg<256, 256, 64, 4, 1>>( a, stream );
		case 7: return launchTiledT<EPI_F32, TileCfg<128, 256, 64, 2, 1>>( a, stream );
		case 8: return launchTiledT<EPI_F32, TileCfg<256, 256, 32, 4, 1>>( a, stream );
#endif
		}
#ifdef WH_PROBES
		setError( "gemm: unknown variant" );
#else
		setError( "gemm: probe variants are not part of this build (WH_PROBES=1 python -m whisper_amd.build --force)" );
#endif
		return -1;
	}

	template<int EPI>
	static int launchSkinnyT( const GemmArgs& a, hipStream_t stream )
	{
		hipLaunchKernelGGL( gemmSkinny<EPI>, dim3( ( a.N + 31 ) / 32 ), dim3( 256 ), 0, stream, a );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int gemmInit() { return 0; }

	static int checkArgs( const GemmArgs& a )
	{
		if( a.M <= 0 || a.N <= 0 || a.K <= 0 || ( a.K % 64 ) != 0 )
		{
			setError( "gemm: M, N must be positive and K a positive multiple of 64" );
			return -1;
		}
		if( ( a.lda % 8 ) != 0 || ( a.aBatchStride % 8 ) != 0 )
		{
			setError( "gemm: A rows must be 16-byte aligned" );
			return -1;
		}
		return 0;
	}

	int launchGemm( const GemmArgs& a, hipStream_t stream )
	{
		WH_CHECK( checkArgs( a ) );
		// big tiles only when they still give every CU a workgroup and M is several clips deep
		const bool big = (long long)( ( a.M + 255 ) / 256 ) * ( ( a.N + 255 ) / 256 ) >= 300 && a.M >= g_opt.gemmBigMinRows && ( g_tuning & TUNE_GEMM_BIG );
		const bool gl = ( g_tuning & TUNE_GEMM_GL ) != 0;
		const bool pf = gl && ( g_tuning & TUNE_GEMM_FRAGPF ) != 0;
		// gemmTiled8 addresses its operands as a 64-bit base + 32-bit byte offsets
		const long long aBytes = 2ll * ( a.Mb > 0 && a.Mb < a.M ? ( (long long)( a.M / a.Mb ) + 1 ) * a.aBatchStride + (long long)a.Mb * a.lda : (long long)a.M * a.lda ) + 2ll * a.K;
		const bool fits32 = aBytes < ( 1ll << 32 ) && 2ll * a.N * a.K < ( 1ll << 32 );
		const bool w8 = big && fits32 && ( g_tuning & TUNE_GEMM_8WAVE ) != 0;
		// gemmTiled4 on top: at least two K tiles, A segments of at least a tile's 256 rows with a non-negative gap
		// WH_GEMM_4WAVE_EPIS: bit mask of epilogues that take gemmTiled4 without the tuning bit (A/B runs)
		static const int epis4 = []() { const char* e = getenv( "WH_GEMM_4WAVE_EPIS" ); return e ? atoi( e ) : 0; }();
		const bool w4 = w8 && a.K >= 256 && ( ( g_tuning & TUNE_GEMM_4WAVE ) != 0 || ( ( epis4 >> a.epi ) & 1 ) != 0 ) &&	   // (K >= 256: the FP16 epilogues leave under the next tile's first four K tiles)
			( a.Mb <= 0 || a.Mb >= a.M || ( a.Mb >= 256 && a.aBatchStride >= (long long)a.Mb * a.lda ) );
#define WH_TILED( E )                                                    \
	if( w4 ) return launchTiled4<E>( a, stream );                        \
	if( w8 && g_opt.gemmMf16 == 1 ) return launchTiled8<E, true>( a, stream ); \
	if( w8 ) return launchTiled8<E>( a, stream );                        \
	if( pf && big ) return launchTiledT<E, CfgGlBigPf>( a, stream );     \
	if( pf ) return launchTiledT<E, CfgGlPf>( a, stream );               \
	if( gl && big ) return launchTiledT<E, CfgGlBig>( a, stream );       \
	if( gl ) return launchTiledT<E, CfgGl>( a, stream );                 \
	if( big ) return launchTiledT<E, CfgBig>( a, stream );               \
	return launchTiledT<E>( a, stream );
		switch( a.epi )
		{
		case EPI_F32: WH_TILED( EPI_F32 )
		case EPI_F16_GELU: WH_TILED( EPI_F16_GELU )
		case EPI_CONV2: if( pf ) return launchTiledT<EPI_CONV2, CfgGlPf>( a, stream ); if( gl ) return launchTiledT<EPI_CONV2, CfgGl>( a, stream ); return launchTiledT<EPI_CONV2>( a, stream );
		case EPI_QKV_ENC: WH_TILED( EPI_QKV_ENC )
		case EPI_CROSS_KV: WH_TILED( EPI_CROSS_KV )
		case EPI_QKV_DEC: if( gl ) return launchTiledT<EPI_QKV_DEC, CfgGl>( a, stream ); return launchTiledT<EPI_QKV_DEC>( a, stream );
		case EPI_Q_DEC: if( gl ) return launchTiledT<EPI_Q_DEC, CfgGl>( a, stream ); return launchTiledT<EPI_Q_DEC>( a, stream );
		}
#undef WH_TILED
		setError( "gemm: unknown epilogue" );
		return -1;
	}

	int launchGemmSkinny( const GemmArgs& a, hipStream_t stream )
	{
		if( a.M > 32 )
			return launchGemm( a, stream );
		WH_CHECK( checkArgs( a ) );
		switch( a.epi )
		{
		case EPI_F32: return launchSkinnyT<EPI_F32>( a, stream );
		case EPI_F16_GELU: return launchSkinnyT<EPI_F16_GELU>( a, stream );
		case EPI_QKV_DEC: return launchSkinnyT<EPI_QKV_DEC>( a, stream );
		case EPI_Q_DEC: return launchSkinnyT<EPI_Q_DEC>( a, stream );
		}
		setError( "gemm: epilogue not available in the skinny kernel" );
		return -1;
	}
}
