// Shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>

namespace wh
{
	typedef _Float16 f16;
	typedef __attribute__( ( ext_vector_type( 8 ) ) ) _Float16 f16x8;
	typedef __attribute__( ( ext_vector_type( 4 ) ) ) _Float16 f16x4;
	typedef __attribute__( ( ext_vector_type( 2 ) ) ) _Float16 f16x2;
	typedef __attribute__( ( ext_vector_type( 16 ) ) ) float f32x16;
	typedef __attribute__( ( ext_vector_type( 4 ) ) ) float f32x4;
	typedef __attribute__( ( ext_vector_type( 4 ) ) ) unsigned int u32x4;
	typedef __attribute__( ( ext_vector_type( 2 ) ) ) unsigned int u32x2;

	constexpr int HEAD_DIM = 64;	// every Whisper size has d/H == 64

	// ---- numerics shared with the reference CPU path (Whisper/source/ggml.c) ----

	// FP32 -> FP16 round-to-nearest-even -> FP32 (ggml.c:150-160, F16C semantics)
	__device__ __forceinline__ float round16( float x ) { return (float)(f16)x; }

	// table_gelu_f16[ fp16(x) ] (ggml.c:1003-1021, :1381): fp16( gelu( fp32( fp16(x) ) ) ) with the tanh form
	__device__ __forceinline__ f16 gelu16( float x )
	{
		// 0.5 f (1 + tanh u) == f / (1 + exp(-2u)); the second form has no cancellation for negative f, so FP32
		// evaluation stays within ~1e-6 relative of the reference's double evaluation before the FP16 rounding.
		// exp(-2u) = 2^( f (C1 + C2 f^2) ): f^2 is exact in FP32 (f has 11 significant bits), one fma, one product, then the
		// hardware's 2^x and reciprocal (1 ulp each). Nine instructions per element instead of ~30 for expf + an IEEE division:
		// the encoder's MLP up-projection spends a third of its time in this epilogue otherwise. Checked against all 63488 finite
		// inputs of the reference's table (tests/test_gpu_ops.py::test_gelu_table_exhaustive: at most 1 ulp, < 0.2 % of the entries).
		const float f = round16( x );
		constexpr float C1 = -2.0f * 0.79788456080286535587989211986876f * 1.44269504088896340736f;
		constexpr float C2 = C1 * 0.044715f;
		const float e = __builtin_amdgcn_exp2f( fmaf( f * f, C2, C1 ) * f );
		const float y = f * __builtin_amdgcn_rcpf( 1.0f + e );
		return (f16)y;
	}

	// table_exp_f16[ fp16(x) ] (ggml.c:1382): fp16( exp( fp32( fp16(x) ) ) ), returned as FP32.
	// exp(f) = 2^(f*log2e): the product is split into its rounded value and the exact remainder (one fma recovers it,
	// a second adds log2e's own rounding error), v_exp_f32 takes the rounded part and the remainder is applied as
	// 2^lo ~ 1 + lo*ln2. Error ~1 ulp of FP32 before the FP16 rounding, i.e. the table value except in ~1e-4 of the
	// inputs where the two round across an FP16 boundary (tests/test_gpu_ops.py checks all 32768 non-positive inputs).
	__device__ __forceinline__ float exp16( float x )
	{
		const float f = round16( x );
		const float L2E = 1.44269502162933349609375f;		// (float)log2(e)
		const float L2E_LO = 1.925963033500163e-08f;		// log2(e) - (float)log2(e)
		const float hi = f * L2E;
		float lo = fmaf( f, L2E, -hi );
		lo = fmaf( f, L2E_LO, lo );
		float r = __builtin_amdgcn_exp2f( hi );
		r = fmaf( r, lo * 0.693147182464599609375f, r );
		return (float)(f16)r;
	}

	__device__ __forceinline__ float waveReduceMax( float v )
	{
#pragma unroll
		for( int o = 32; o > 0; o >>= 1 )
			v = fmaxf( v, __shfl_xor( v, o, 64 ) );
		return v;
	}
	__device__ __forceinline__ float waveReduceSum( float v )
	{
#pragma unroll
		for( int o = 32; o > 0; o >>= 1 )
			v += __shfl_xor( v, o, 64 );
		return v;
	}
	__device__ __forceinline__ double waveReduceSumD( double v )
	{
#pragma unroll
		for( int o = 32; o > 0; o >>= 1 )
			v += __shfl_xor( v, o, 64 );
		return v;
	}

	// LayerNorm + affine for NR rows by one wavefront: the numerics of ggml_compute_forward_norm_f32 followed by w*y + b
	// (ggml.c:4098-4156, whisper.cpp:1195-1199), FP16 result (the consumer is always a GEMM that rounds its activations).
	// The reference sums in double; FP32 two-pass with a wavefront shuffle tree differs by ~1e-7 relative.
	// Rows are xr + j * rowStride, j < NR; rows at or beyond nRows are skipped. A lane owns 4 consecutive columns of every
	// 256-column chunk, so x, w and b arrive as 16-byte loads, all requested before the first use, and the NR reduction
	// chains are interleaved. d is a multiple of 4, at most 256 * MAXC. store( j, c, f16x4 ).
	// Every rounding is spelled out (no contraction), so a row gets the same bits from every instantiation and caller:
	// the standalone kernel, the gemv prologue, any NR.
	template<int MAXC, int NR, class Store>
	__device__ __forceinline__ void layerNormRows( const float* __restrict__ xr, long long rowStride, int nRows, const float* __restrict__ w,
		const float* __restrict__ b, int d, int lane, Store&& store )
	{
		f32x4 v[ NR ][ MAXC ], wv[ MAXC ], bv[ MAXC ];
#pragma unroll
		for( int i = 0; i < MAXC; i++ )
		{
			int c = ( lane + 64 * i ) * 4;
			c = c < d ? c : d - 4;
#pragma unroll
			for( int j = 0; j < NR; j++ )
			{
				const int jr = j < nRows ? j : ( nRows > 0 ? nRows - 1 : 0 );
				v[ j ][ i ] = *(const f32x4*)( xr + jr * rowStride + c );
			}
			wv[ i ] = *(const f32x4*)( w + c );
			bv[ i ] = *(const f32x4*)( b + c );
		}
		const float invD = 1.0f / (float)d;
		float s[ NR ];
#pragma unroll
		for( int j = 0; j < NR; j++ )
		{
			s[ j ] = 0.0f;
#pragma unroll
			for( int i = 0; i < MAXC; i++ )
				if( ( lane + 64 * i ) * 4 < d )
					s[ j ] = __fadd_rn( s[ j ], __fadd_rn( __fadd_rn( v[ j ][ i ][ 0 ], v[ j ][ i ][ 1 ] ), __fadd_rn( v[ j ][ i ][ 2 ], v[ j ][ i ][ 3 ] ) ) );
		}
#pragma unroll
		for( int o = 32; o > 0; o >>= 1 )
#pragma unroll
			for( int j = 0; j < NR; j++ ) s[ j ] = __fadd_rn( s[ j ], __shfl_xor( s[ j ], o, 64 ) );
		float s2[ NR ];
#pragma unroll
		for( int j = 0; j < NR; j++ )
		{
			const float mean = __fmul_rn( s[ j ], invD );
			s2[ j ] = 0.0f;
#pragma unroll
			for( int i = 0; i < MAXC; i++ )
			{
#pragma unroll
				for( int e = 0; e < 4; e++ ) v[ j ][ i ][ e ] = __fsub_rn( v[ j ][ i ][ e ], mean );
				if( ( lane + 64 * i ) * 4 < d )
				{
					float t = __fmul_rn( v[ j ][ i ][ 0 ], v[ j ][ i ][ 0 ] );
					t = fmaf( v[ j ][ i ][ 1 ], v[ j ][ i ][ 1 ], t );
					t = fmaf( v[ j ][ i ][ 2 ], v[ j ][ i ][ 2 ], t );
					t = fmaf( v[ j ][ i ][ 3 ], v[ j ][ i ][ 3 ], t );
					s2[ j ] = __fadd_rn( s2[ j ], t );
				}
			}
		}
#pragma unroll
		for( int o = 32; o > 0; o >>= 1 )
#pragma unroll
			for( int j = 0; j < NR; j++ ) s2[ j ] = __fadd_rn( s2[ j ], __shfl_xor( s2[ j ], o, 64 ) );
#pragma unroll
		for( int j = 0; j < NR; j++ )
		{
			if( j >= nRows ) continue;
			const float scale = 1.0f / sqrtf( __fadd_rn( __fmul_rn( s2[ j ], invD ), 1e-5f ) );
#pragma unroll
			for( int i = 0; i < MAXC; i++ )
			{
				const int c = ( lane + 64 * i ) * 4;
				if( c < d )
				{
					f16x4 h;
#pragma unroll
					for( int e = 0; e < 4; e++ )
						h[ e ] = (f16)__fadd_rn( __fmul_rn( __fmul_rn( v[ j ][ i ][ e ], scale ), wv[ i ][ e ] ), bv[ i ][ e ] );
					store( j, c, h );
				}
			}
		}
	}

	// ---- host side ----
	// hipFuncSetAttribute is a per-DEVICE setting: one bit per device ordinal remembers where it has been applied, so a
	// model on adapter N > 0 gets its dynamic-LDS limit too (the reference binds one D3D device per model,
	// Whisper/ML/Device.cpp:163-177). The object is a function-local static shared by every host thread, so it keeps NO
	// per-call state: needed() hands the calling thread's device ordinal back to the caller (or -1 when the attribute is
	// already set there) and mark( ordinal ) records exactly that device. Two threads on different adapters can interleave
	// freely; two threads on the same adapter may both set the attribute, which is idempotent.
	struct PerDeviceOnce
	{
		std::atomic<unsigned long long> done{ 0 };
		int needed() const
		{
			int device = 0;
			if( hipGetDevice( &device ) != hipSuccess ) device = 0;
			device &= 63;
			return ( ( done.load( std::memory_order_acquire ) >> device ) & 1ull ) == 0 ? device : -1;
		}
		void mark( int device ) { done.fetch_or( 1ull << ( device & 63 ), std::memory_order_release ); }
	};
	void setError( const std::string& s );
	int hipFail( hipError_t e, const char* what, const char* file, int line );
}

#define WH_HIP( expr )                                                              \
	do                                                                              \
	{                                                                               \
		hipError_t e__ = ( expr );                                                  \
		if( e__ != hipSuccess ) return wh::hipFail( e__, #expr, __FILE__, __LINE__ ); \
	} while( 0 )

#define WH_CHECK( st )               \
	do                               \
	{                                \
		const int s__ = ( st );      \
		if( s__ != 0 ) return s__;   \
	} while( 0 )
