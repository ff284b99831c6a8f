// Does the ORDER in which a wave's MFMAs walk its accumulator tiles change what the chip can sustain on random data? One wave per SIMD (1024 waves), 256 accumulator
// registers per wave as in a 128 x 128 wave tile, 8 (16x16x32: NA = NB = 8) or 4 (32x32x16: NA = NB = 4) A and B fragments in registers, nothing but MFMAs in the loop.
//   order 0: A-stationary rows   for i: for j: acc[i][j] += A[i] B[j]     (one operand unchanged over NB consecutive instructions)
//   order 1: diagonal            both operands change with every instruction
//   order 2: B-stationary columns
// hipcc --offload-arch=gfx950 -O2 tools/mfma_order_probe.hip -o whisper_amd/lib/mfma-order-probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
typedef _Float16 f16;
typedef __attribute__( ( ext_vector_type( 8 ) ) ) _Float16 f16x8;
typedef __attribute__( ( ext_vector_type( 16 ) ) ) float f32x16;
typedef __attribute__( ( ext_vector_type( 4 ) ) ) float f32x4;

template<int SHAPE, int ORDER>
__global__ void __launch_bounds__( 256, 1 ) k( const f16x8* in, float* out, int iters )
{
	constexpr int NF = SHAPE == 16 ? 8 : 4;
	const int t = blockIdx.x * 256 + threadIdx.x;
	f16x8 a[ NF ], b[ NF ];
	for( int i = 0; i < NF; i++ )
	{
		a[ i ] = in[ ( t * 16 + i ) & 0xffff ];
		b[ i ] = in[ ( t * 16 + 8 + i ) & 0xffff ];
	}
	using Acc = typename std::conditional<SHAPE == 16, f32x4, f32x16>::type;
	constexpr int NR = SHAPE == 16 ? 4 : 16;
	Acc c[ NF ][ NF ];
	for( int i = 0; i < NF; i++ )
		for( int j = 0; j < NF; j++ )
			for( int r = 0; r < NR; r++ ) c[ i ][ j ][ r ] = 0.0f;
	for( int it = 0; it < iters; it++ )
	{
#pragma unroll
		for( int s = 0; s < NF * NF; s++ )
		{
			const int i = ORDER == 0 ? s / NF : ORDER == 1 ? s % NF : s % NF;
			const int j = ORDER == 0 ? s % NF : ORDER == 1 ? ( s + s / NF ) % NF : s / NF;
			if constexpr( SHAPE == 16 )
				c[ i ][ j ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( a[ i ], b[ j ], c[ i ][ j ], 0, 0, 0 );
			else
				c[ i ][ j ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( a[ i ], b[ j ], c[ i ][ j ], 0, 0, 0 );
		}
		// keep the fragments from being folded: rotate them (cheap, once per NF * NF MFMAs)
		asm volatile( "" : "+v"( a[ 0 ] ), "+v"( b[ 0 ] ) );
	}
	float s = 0;
	for( int i = 0; i < NF; i++ )
		for( int j = 0; j < NF; j++ )
			for( int r = 0; r < NR; r++ ) s += c[ i ][ j ][ r ];
	out[ t ] = s;
}


// gemmTiled8's quadrant with 16x16x32: 4 row tiles x 2 column tiles x 2 k-halves = 16 MFMAs per quadrant, two quadrants per iteration (16 accumulator tiles).
//   order 0: srcA fixed for 2 (for h: for i: for j)   order 1: srcB fixed for 4 (for h: for j: for i)   order 2: both change (i, j diagonal)
//   order 3: operand roles swapped -- the B fragment as srcA, fixed for 4 (computes the transposed tile)
template<int ORDER>
__global__ void __launch_bounds__( 256, 1 ) kq( const f16x8* in, float* out, int iters )
{
	const int t = blockIdx.x * 256 + threadIdx.x;
	f16x8 a[ 2 ][ 4 ][ 2 ], b[ 2 ][ 2 ][ 2 ];
	int n = 0;
	for( int q = 0; q < 2; q++ )
		for( int h = 0; h < 2; h++ )
		{
			for( int i = 0; i < 4; i++ ) a[ q ][ i ][ h ] = in[ ( t * 32 + n++ ) & 0xffff ];
			for( int j = 0; j < 2; j++ ) b[ q ][ j ][ h ] = in[ ( t * 32 + n++ ) & 0xffff ];
		}
	f32x4 c[ 2 ][ 4 ][ 2 ];
	for( int q = 0; q < 2; q++ )
		for( int i = 0; i < 4; i++ )
			for( int j = 0; j < 2; j++ )
				for( int r = 0; r < 4; r++ ) c[ q ][ i ][ j ][ r ] = 0.0f;
	for( int it = 0; it < iters; it++ )
	{
#pragma unroll
		for( int q = 0; q < 2; q++ )
#pragma unroll
			for( int h = 0; h < 2; h++ )
#pragma unroll
				for( int s = 0; s < 8; s++ )
				{
					const int i = ORDER == 0 ? s / 2 : ORDER == 2 ? s % 4 : s % 4;
					const int j = ORDER == 0 ? s % 2 : ORDER == 2 ? ( s + s / 4 + ( s & 1 ) ) % 2 : s / 4;
					if constexpr( ORDER == 3 )
						c[ q ][ i ][ j ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( b[ q ][ j ][ h ], a[ q ][ i ][ h ], c[ q ][ i ][ j ], 0, 0, 0 );
					else
						c[ q ][ i ][ j ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( a[ q ][ i ][ h ], b[ q ][ j ][ h ], c[ q ][ i ][ j ], 0, 0, 0 );
				}
		asm volatile( "" : "+v"( a[ 0 ][ 0 ][ 0 ] ), "+v"( b[ 0 ][ 0 ][ 0 ] ) );
	}
	float s = 0;
	for( int q = 0; q < 2; q++ )
		for( int i = 0; i < 4; i++ )
			for( int j = 0; j < 2; j++ )
				for( int r = 0; r < 4; r++ ) s += c[ q ][ i ][ j ][ r ];
	out[ t ] = s;
}

template<int ORDER>
static double runq( const f16x8* in, float* out, int iters, int wgs )
{
	hipEvent_t e0, e1;
	(void)hipEventCreate( &e0 ); (void)hipEventCreate( &e1 );
	hipLaunchKernelGGL( ( kq<ORDER> ), dim3( wgs ), dim3( 256 ), 0, 0, in, out, iters / 10 );
	(void)hipDeviceSynchronize();
	(void)hipEventRecord( e0, 0 );
	hipLaunchKernelGGL( ( kq<ORDER> ), dim3( wgs ), dim3( 256 ), 0, 0, in, out, iters );
	(void)hipEventRecord( e1, 0 );
	(void)hipEventSynchronize( e1 );
	float ms = 0;
	(void)hipEventElapsedTime( &ms, e0, e1 );
	return 32.0 * 2 * 16 * 16 * 32 * iters * 4.0 * wgs / ( ms * 1e-3 ) / 1e12;
}

template<int SHAPE, int ORDER>
static double run( const f16x8* in, float* out, int iters )
{
	hipEvent_t e0, e1;
	(void)hipEventCreate( &e0 ); (void)hipEventCreate( &e1 );
	hipLaunchKernelGGL( ( k<SHAPE, ORDER> ), dim3( 256 ), dim3( 256 ), 0, 0, in, out, iters / 10 );
	(void)hipDeviceSynchronize();
	(void)hipEventRecord( e0, 0 );
	hipLaunchKernelGGL( ( k<SHAPE, ORDER> ), dim3( 256 ), dim3( 256 ), 0, 0, in, out, iters );
	(void)hipEventRecord( e1, 0 );
	(void)hipEventSynchronize( e1 );
	float ms = 0;
	(void)hipEventElapsedTime( &ms, e0, e1 );
	const double perIter = SHAPE == 16 ? 64.0 * 2 * 16 * 16 * 32 : 16.0 * 2 * 32 * 32 * 16;
	return perIter * iters * 1024.0 / ( ms * 1e-3 ) / 1e12;
}

int main()
{
	const int n = 65536;
	f16x8* h = (f16x8*)malloc( n * 16 );
	f16x8 *dRand, *dZero; float* out;
	(void)hipMalloc( &dRand, n * 16 ); (void)hipMalloc( &dZero, n * 16 ); (void)hipMalloc( &out, 1024 * 256 * 4 );
	uint32_t seed = 1;
	for( int i = 0; i < n; i++ )
		for( int j = 0; j < 8; j++ )
		{
			seed = seed * 1664525u + 1013904223u;
			h[ i ][ j ] = (f16)( ( (int)( seed >> 16 ) - 32768 ) / 65536.0f );
		}
	(void)hipMemcpy( dRand, h, n * 16, hipMemcpyHostToDevice );
	(void)hipMemset( dZero, 0, n * 16 );
	const int iters = 40000;
	for( int rep = 0; rep < 3; rep++ )
	{
		printf( "16x16x32 random: A-stationary %.0f | diagonal %.0f | B-stationary %.0f TF      32x32x16 random: A-stationary %.0f | diagonal %.0f | B-stationary %.0f TF      zeros: %.0f / %.0f\n",
			run<16, 0>( dRand, out, iters ), run<16, 1>( dRand, out, iters ), run<16, 2>( dRand, out, iters ),
			run<32, 0>( dRand, out, iters ), run<32, 1>( dRand, out, iters ), run<32, 2>( dRand, out, iters ),
			run<16, 0>( dZero, out, iters ), run<32, 0>( dZero, out, iters ) );
		fflush( stdout );
	}
	for( int wgs = 256; wgs <= 512; wgs *= 2 )
		for( int rep = 0; rep < 2; rep++ )
		{
			printf( "quadrant 4 x 2 x 2, 16x16x32, random, %d waves: srcA fixed for 2 %.0f | srcB fixed for 4 %.0f | both change %.0f | roles swapped, srcA fixed for 4 %.0f TF\n", 4 * wgs,
				runq<0>( dRand, out, iters * 2, wgs ), runq<1>( dRand, out, iters * 2, wgs ), runq<2>( dRand, out, iters * 2, wgs ), runq<3>( dRand, out, iters * 2, wgs ) );
			fflush( stdout );
		}
	return 0;
}
