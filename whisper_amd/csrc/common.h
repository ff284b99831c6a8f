// Shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
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
		const float f = round16( x );
		const float u2 = -2.0f * 0.79788456080286535587989211986876f * f * ( 1.0f + 0.044715f * f * f );
		const float y = f / ( 1.0f + expf( u2 ) );
		return (f16)y;
	}

	// table_exp_f16[ fp16(x) ] (ggml.c:1382): fp16( exp( fp32( fp16(x) ) ) ), returned as FP32
	__device__ __forceinline__ float exp16( float x )
	{
		const float f = round16( x );
		return (float)(f16)expf( f );
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

	// ---- host side ----
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
