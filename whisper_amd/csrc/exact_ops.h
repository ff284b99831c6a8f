// The reference CPU path's arithmetic, operation for operation and in ITS summation order -- the primitives of WH_FLAG_PARITY_EXACT.
//
// Whisper/source/ggml.c (the code behind eModelImplementation::Reference, built with AVX2 + FMA + F16C: SURVEY.md section 8(c)) fixes
// an order for every sum it takes. These functions restate that order with scalar IEEE operations, so that a kernel built from them
// produces the reference's bits, not values "within a tolerance" of them:
//   * ggml_vec_dot_f16 (ggml.c:751-790, macros :452-528): 32 interleaved FP32 chains (element i feeds chain i mod 32: four 8-lane
//     accumulators), each a sequence of fused multiply-adds; then the fixed tree of GGML_F32x8_REDUCE; elements beyond the last
//     multiple of 32 are added one by one in DOUBLE (their products are exact) and the total is rounded to FP32 once.
//   * ggml_vec_mad_f16 (ggml.c:871-891): y = fp16( fma( fp32(x), v, fp32(y) ) ), the FP16-accumulated P.V of the decoder.
//   * ggml_compute_forward_norm_f32 (ggml.c:4098-4156): sequential double sums; the second one is a fused multiply-add because gcc
//     contracts `sum2 += v*v` under -mfma (checked in the disassembly of oracle/_ref/ggml.o: vfmadd231sd).
//   * soft_max / flash_attn_f16's softmax (ggml.c:5026-5096, :6036-6063): max, table_exp_f16[ fp16( s - max ) ], a double sum (exact in
//     any order: at most 2^16 multiples of 2^-24 below 2), one FP32 reciprocal factor.
// Everything is __host__ __device__ and free of HIP types so that tests/exact_cpu compiles the same source with g++ and holds it
// against oracle/_ref on the CPU (tests/test_exact_cpu.py); the kernels are in exact.hip. This file must be compiled with
// -ffp-contract=off: every fused operation is spelled fma()/fmaf(), every other product and sum rounds on its own.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined( __HIPCC__ )
#define WH_HD __host__ __device__ __forceinline__
#else
#define WH_HD inline
#endif

namespace whx
{
	typedef _Float16 h16;

	WH_HD float toF32( h16 v ) { return (float)v; }
	// FP32 -> FP16, round to nearest even, like _cvtss_sh( x, 0 ) (ggml.c:159) -- of a value that HAS BEEN ROUNDED TO FP32. On gfx950 the compiler
	// selects v_fma_mixlo_f16 for fptrunc( fmul / fma ), and that instruction rounds the exact product ONCE, to FP16: the reference rounds twice
	// (F32 result, then F16C), and the two differ in ~2^-13 of the cases (measured: 443 of 192000 outputs of one encoder layer's attention). The empty
	// asm makes the FP32 value opaque, so the conversion stays a v_cvt_f16_f32 of the rounded FP32 result.
	WH_HD h16 toF16( float v )
	{
#if defined( __HIP_DEVICE_COMPILE__ )
		asm volatile( "" : "+v"( v ) );
#endif
		return (h16)v;
	}
	WH_HD uint16_t bitsOf( h16 v )
	{
		union { h16 h; uint16_t u; } c;
		c.h = v;
		return c.u;
	}

	// ---- ggml_vec_dot_f16 ----
	struct Dot16
	{
		float a[ 32 ];
		WH_HD void clear()
		{
#pragma unroll
			for( int i = 0; i < 32; i++ ) a[ i ] = 0.0f;
		}
		// one GGML_F16_STEP: elements [0, 32) of both operands
		WH_HD void step( const h16* x, const h16* y )
		{
#pragma unroll
			for( int i = 0; i < 32; i++ ) a[ i ] = fmaf( toF32( x[ i ] ), toF32( y[ i ] ), a[ i ] );
		}
		WH_HD void step1( int lane, float x, float y ) { a[ lane ] = fmaf( x, y, a[ lane ] ); }
		// GGML_F32x8_REDUCE (ggml.c:472-487)
		WH_HD float reduce() const
		{
			float r[ 8 ];
#pragma unroll
			for( int e = 0; e < 8; e++ ) r[ e ] = ( a[ e ] + a[ e + 8 ] ) + ( a[ e + 16 ] + a[ e + 24 ] );
			const float t0 = r[ 0 ] + r[ 4 ], t1 = r[ 1 ] + r[ 5 ], t2 = r[ 2 ] + r[ 6 ], t3 = r[ 3 ] + r[ 7 ];
			return ( t0 + t1 ) + ( t2 + t3 );
		}
	};

	// NOT the reference's order: the same 32 chains added left to right. Exists to measure how far the smallest change of summation order moves the
	// logits (tools/parity_split.py, "exact_alt_order"): the yardstick for any implementation that does not sum in ggml's order.
	WH_HD float reduceLeftToRight( const Dot16& d )
	{
		float s = d.a[ 0 ];
#pragma unroll
		for( int i = 1; i < 32; i++ ) s = s + d.a[ i ];
		return s;
	}

	// The whole of ggml_vec_dot_f16 for contiguous operands of any length (the leftovers in double, ggml.c:783-786).
	WH_HD float dot16( const h16* x, const h16* y, int n )
	{
		Dot16 d;
		d.clear();
		const int np = n & ~31;
		for( int i = 0; i < np; i += 32 ) d.step( x + i, y + i );
		double sumf = (double)d.reduce();
		for( int i = np; i < n; i++ ) sumf += (double)( toF32( x[ i ] ) * toF32( y[ i ] ) );
		return (float)sumf;
	}

	// ---- ggml_vec_mad_f16, one element ----
	WH_HD h16 mad16( h16 y, h16 x, float v ) { return toF16( fmaf( toF32( x ), v, toF32( y ) ) ); }

	// ---- ggml_compute_forward_norm_f32, one row, followed by w * y + b as two operations (whisper.cpp:1190-1199) ----
	// x, w, b, out: n contiguous floats (out may alias x).
	WH_HD void normRow( const float* x, const float* w, const float* b, float* out, int n )
	{
		double mean = 0.0;
		for( int i = 0; i < n; i++ ) mean += (double)x[ i ];
		mean /= (double)n;
		double sum2 = 0.0;
		for( int i = 0; i < n; i++ )
		{
			const double v = (double)x[ i ] - mean;
			sum2 = fma( v, v, sum2 );
		}
		const double eps = (double)1e-5f;
		const float scale = (float)( 1.0 / sqrt( sum2 / (double)n + eps ) );
		for( int i = 0; i < n; i++ )
		{
			const float y = (float)( (double)x[ i ] - mean );
			const float s = y * scale;
			const float t = w[ i ] * s;
			out[ i ] = t + b[ i ];
		}
	}

	// ---- table lookups (ggml.c:1006-1021, :5069-5080): tables are the 65536-entry FP16 tables ggml_init builds ----
	WH_HD float gelu16( const h16* table, float x ) { return toF32( table[ bitsOf( toF16( x ) ) ] ); }
	WH_HD float exp16( const h16* table, float x ) { return toF32( table[ bitsOf( toF16( x ) ) ] ); }
}	// namespace whx
