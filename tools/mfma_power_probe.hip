// What clock does the chip hold under back-to-back MFMAs on random FP16 data, for the two FP16 shapes? (nothing but MFMAs: 4 waves per CU, one per
// SIMD, 8 independent accumulator chains per wave; operands random / zero). TFLOP/s here = matrix-pipe rate x clock: the ratio to 2.5 PF @ 2.4 GHz is the clock.
// hipcc --offload-arch=gfx950 -O2 tools/mfma_power_probe.hip -o whisper_amd/lib/mfma-power-probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <type_traits>
typedef _Float16 f16;
typedef __attribute__( ( ext_vector_type( 8 ) ) ) _Float16 f16x8;
typedef __attribute__( ( ext_vector_type( 16 ) ) ) float f32x16;
typedef __attribute__( ( ext_vector_type( 4 ) ) ) float f32x4;

template<int SHAPE>
__global__ void __launch_bounds__( 256 ) k( const f16x8* in, float* out, int iters )
{
	const int t = blockIdx.x * 256 + threadIdx.x;
	f16x8 a[ 4 ], b[ 4 ];
	for( int i = 0; i < 4; i++ )
	{
		a[ i ] = in[ ( t * 8 + i ) & 0xffff ];
		b[ i ] = in[ ( t * 8 + 4 + i ) & 0xffff ];
	}
	if constexpr( SHAPE == 32 )
	{
		f32x16 c[ 8 ];
		for( int i = 0; i < 8; i++ )
			for( int r = 0; r < 16; r++ ) c[ i ][ r ] = 0.0f;
		for( int it = 0; it < iters; it++ )
#pragma unroll
			for( int i = 0; i < 8; i++ ) c[ i ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( a[ i & 3 ], b[ ( i >> 1 ) & 3 ], c[ i ], 0, 0, 0 );
		float s = 0;
		for( int i = 0; i < 8; i++ )
			for( int r = 0; r < 16; r++ ) s += c[ i ][ r ];
		out[ t ] = s;
	}
	else
	{
		f32x4 c[ 16 ];
		for( int i = 0; i < 16; i++ )
			for( int r = 0; r < 4; r++ ) c[ i ][ r ] = 0.0f;
		for( int it = 0; it < iters; it++ )
#pragma unroll
			for( int i = 0; i < 16; i++ ) c[ i ] = __builtin_amdgcn_mfma_f32_16x16x32_f16( a[ i & 3 ], b[ ( i >> 2 ) & 3 ], c[ i ], 0, 0, 0 );
		float s = 0;
		for( int i = 0; i < 16; i++ )
			for( int r = 0; r < 4; r++ ) s += c[ i ][ r ];
		out[ t ] = s;
	}
}

// The same MFMA stream with the OTHER traffic of a GEMM K loop beside it, at gemmTiled4's ratio per 16 MFMAs: 4 LDS-DMA pieces of 1 KiB from an L2-resident
// buffer (MODE & 1) and / or 8 ds_read_b128 fragment reads (MODE & 2). Two workgroups per CU (two waves per SIMD) so that issue stalls of one wave are covered by the
// other and a lower rate means a lower CLOCK (power), not an idle matrix pipe.
template<int MODE>
__global__ void __launch_bounds__( 256, 2 ) kMix( const f16x8* in, const unsigned char* l2buf, unsigned l2mask, float* out, int iters )
{
	extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char lds[];	  // 64 KiB
	typedef __attribute__( ( address_space( 3 ) ) ) void* LdsPtr;
	const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane( threadIdx.x >> 6 );
	f16x8 a[ 4 ], b[ 4 ];
	for( int i = 0; i < 4; i++ )
	{
		a[ i ] = in[ ( t * 8 + i ) & 0xffff ];
		b[ i ] = in[ ( t * 8 + 4 + i ) & 0xffff ];
	}
	for( int i = threadIdx.x; i < 4096; i += 256 ) ( (f16x8*)lds )[ i ] = in[ ( i + blockIdx.x * 17 ) & 0xffff ];
	__syncthreads();
	f32x16 c[ 8 ];
	for( int i = 0; i < 8; i++ )
		for( int r = 0; r < 16; r++ ) c[ i ][ r ] = 0.0f;
	const unsigned ldsBase = __builtin_amdgcn_readfirstlane( (unsigned)(size_t)(LdsPtr)lds ) + wave * 16384;
	// MODE & 8: the 16-byte chunks of every 128-byte row fetched in a permuted order (chunk ^ row, the encoder kernels' swizzled source) instead of lane order
	const unsigned laneSrc = ( MODE & 8 ) ? ( ( lane & ~7u ) | ( ( lane & 7u ) ^ ( ( lane >> 3 ) & 7u ) ) ) : (unsigned)lane;
	unsigned off = ( blockIdx.x * 65536u + wave * 16384u + laneSrc * 16u ) & l2mask;
	f16x8 fr[ 8 ];
	for( int i = 0; i < 8; i++ ) fr[ i ] = a[ i & 3 ];
	// MODE & 4: the same 4 KiB per wave and iteration as plain 16-byte loads into registers, written to LDS two iterations later (what a register-staged K loop does)
	f16x8 st[ 2 ][ 4 ];
	if constexpr( ( MODE & 4 ) != 0 )
	{
		for( int q = 0; q < 2; q++ )
		{
			for( int p = 0; p < 4; p++ ) st[ q ][ p ] = *(const f16x8*)( l2buf + ( ( off + p * 1024u ) & l2mask ) );
			off = ( off + 4096u * 61u ) & l2mask;
		}
	}
	auto iteration = [ & ]( int it, auto parc )
	{
		constexpr int PAR = decltype( parc )::value;
		if constexpr( ( MODE & 4 ) != 0 )
		{
#pragma unroll
			for( int p = 0; p < 4; p++ ) *(f16x8*)( lds + wave * 16384 + ( ( it * 4 + p ) & 15 ) * 1024 + lane * 16 ) = st[ PAR ][ p ];
#pragma unroll
			for( int p = 0; p < 4; p++ ) st[ PAR ][ p ] = *(const f16x8*)( l2buf + ( ( off + p * 1024u ) & l2mask ) );
			off = ( off + 4096u * 61u ) & l2mask;
		}
		if constexpr( ( MODE & 1 ) != 0 )
		{
#pragma unroll
			for( int p = 0; p < 4; p++ )
			{
				const unsigned dst = ldsBase + ( ( it * 4 + p ) & 15 ) * 1024;
				const unsigned o = ( off + p * 1024u ) & l2mask;
				unsigned keep;
				asm volatile( "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0" : "=&s"( keep ) : "s"( (const void*)l2buf ), "v"( o ), "s"( dst ) : "memory" );
			}
			off = ( off + 4096u * 61u ) & l2mask;
			asm volatile( "s_waitcnt vmcnt(8)" ::: "memory" );
		}
		if constexpr( ( MODE & 2 ) != 0 )
		{
#pragma unroll
			for( int p = 0; p < 8; p++ ) fr[ p ] = *(const f16x8*)( lds + ( ( ( it * 8 + p ) * 1024 + lane * 16 + wave * 16384 ) & 65535 ) );
		}
#pragma unroll
		for( int i = 0; i < 8; i++ ) c[ i ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( ( MODE & 2 ) ? fr[ i ] : a[ i & 3 ], b[ ( i >> 1 ) & 3 ], c[ i ], 0, 0, 0 );
#pragma unroll
		for( int i = 0; i < 8; i++ ) c[ i ] = __builtin_amdgcn_mfma_f32_32x32x16_f16( a[ i & 3 ], ( MODE & 2 ) ? fr[ 7 - i ] : b[ ( i >> 1 ) & 3 ], c[ i ], 0, 0, 0 );
	};
	for( int it = 0; it < iters; it += 2 )
	{
		iteration( it, std::integral_constant<int, 0>{} );
		iteration( it + 1, std::integral_constant<int, 1>{} );
	}
	asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
	float s = 0;
	for( int i = 0; i < 8; i++ )
		for( int r = 0; r < 16; r++ ) s += c[ i ][ r ];
	out[ t ] = s;
}

template<int MODE>
static double runMix( const f16x8* in, const unsigned char* l2buf, float* out, int iters )
{
	hipEvent_t e0, e1;
	hipEventCreate( &e0 ); hipEventCreate( &e1 );
	hipFuncSetAttribute( (const void*)kMix<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 );
	const unsigned mask = ( 4u << 20 ) - 1;
	hipLaunchKernelGGL( kMix<MODE>, dim3( 512 ), dim3( 256 ), 65536, 0, in, l2buf, mask, out, iters / 10 );
	hipDeviceSynchronize();
	hipEventRecord( e0, 0 );
	hipLaunchKernelGGL( kMix<MODE>, dim3( 512 ), dim3( 256 ), 65536, 0, in, l2buf, mask, out, iters );
	hipEventRecord( e1, 0 );
	hipEventSynchronize( e1 );
	float ms = 0;
	hipEventElapsedTime( &ms, e0, e1 );
	return 16.0 * 2 * 32 * 32 * 16 * iters * 2048.0 / ( ms * 1e-3 ) / 1e12;
}

template<int SHAPE>
static double run( const f16x8* in, float* out, int iters, int wgs = 256 )
{
	hipEvent_t e0, e1;
	hipEventCreate( &e0 ); hipEventCreate( &e1 );
	hipLaunchKernelGGL( k<SHAPE>, dim3( wgs ), dim3( 256 ), 0, 0, in, out, iters / 10 );
	hipDeviceSynchronize();
	hipEventRecord( e0, 0 );
	hipLaunchKernelGGL( k<SHAPE>, dim3( wgs ), dim3( 256 ), 0, 0, in, out, iters );
	hipEventRecord( e1, 0 );
	hipEventSynchronize( e1 );
	float ms = 0;
	hipEventElapsedTime( &ms, e0, e1 );
	const double perIter = SHAPE == 32 ? 8.0 * 2 * 32 * 32 * 16 : 16.0 * 2 * 16 * 16 * 32;
	return perIter * iters * 4.0 * wgs / ( ms * 1e-3 ) / 1e12;	  // 4 waves per workgroup
}

int main()
{
	const int n = 65536;
	f16x8* h = (f16x8*)malloc( n * 16 );
	f16x8 *dRand, *dZero; float* out;
	hipMalloc( &dRand, n * 16 ); hipMalloc( &dZero, n * 16 ); hipMalloc( &out, 1024 * 256 * 4 );
	uint32_t seed = 1;
	for( int i = 0; i < n; i++ )
		for( int j = 0; j < 8; j++ )
		{
			seed = seed * 1664525u + 1013904223u;
			h[ i ][ j ] = (f16)( ( (int)( seed >> 16 ) - 32768 ) / 32768.0f );
		}
	hipMemcpy( dRand, h, n * 16, hipMemcpyHostToDevice );
	hipMemset( dZero, 0, n * 16 );
	const int iters = 200000;
	for( int rep = 0; rep < 2; rep++ )
	{
		const double r32 = run<32>( dRand, out, iters ), r16 = run<16>( dRand, out, iters );
		const double z32 = run<32>( dZero, out, iters ), z16 = run<16>( dZero, out, iters );
		printf( "MFMA only, 1024 waves: 32x32x16 random %.0f TF (%.2f GHz)  16x16x32 random %.0f TF (%.2f GHz)   32x32x16 zeros %.0f TF (%.2f GHz)  16x16x32 zeros %.0f TF (%.2f GHz)\n",
			r32, r32 / 2500 * 2.4, r16, r16 / 2500 * 2.4, z32, z32 / 2500 * 2.4, z16, z16 / 2500 * 2.4 );
	}
	// two and four waves per SIMD (512 / 1024 workgroups): the 16x16x32 form needs more than one wave per SIMD to issue back to back
	for( int wgs = 512; wgs <= 1024; wgs *= 2 )
	{
		const double r32 = run<32>( dRand, out, iters / 2, wgs ), r16 = run<16>( dRand, out, iters / 2, wgs );
		const double z32 = run<32>( dZero, out, iters / 2, wgs ), z16 = run<16>( dZero, out, iters / 2, wgs );
		printf( "MFMA only, %d waves: 32x32x16 random %.0f TF  16x16x32 random %.0f TF   32x32x16 zeros %.0f TF  16x16x32 zeros %.0f TF\n", 4 * wgs, r32, r16, z32, z16 );
	}
	// the other traffic of a K loop beside the MFMAs (random operands, 2048 waves)
	unsigned char* l2buf;
	hipMalloc( &l2buf, 4u << 20 );
	for( int i = 0; i < 64; i++ ) hipMemcpy( l2buf + i * 65536, dRand, 65536, hipMemcpyDeviceToDevice );
	for( int rep = 0; rep < 2; rep++ )
	{
		const double m0 = runMix<0>( dRand, l2buf, out, iters / 4 ), m1 = runMix<1>( dRand, l2buf, out, iters / 4 ), m2 = runMix<2>( dRand, l2buf, out, iters / 4 ), m3 = runMix<3>( dRand, l2buf, out, iters / 4 );
		printf( "16 MFMAs per iteration, 2048 waves, random: alone %.0f TF | + 4 KiB of LDS-DMA from L2 %.0f | + 8 ds_read_b128 %.0f | + both %.0f\n", m0, m1, m2, m3 );
		const double m9 = runMix<9>( dRand, l2buf, out, iters / 4 ), m11 = runMix<11>( dRand, l2buf, out, iters / 4 );
		printf( "   LDS-DMA with the chunks of a row in permuted order: %.0f TF | + 8 ds_read_b128 %.0f\n", m9, m11 );
		const double m4 = runMix<4>( dRand, l2buf, out, iters / 4 ), m6 = runMix<6>( dRand, l2buf, out, iters / 4 );
		printf( "   the same 4 KiB as plain loads to registers + ds_write_b128: %.0f TF | + 8 ds_read_b128 %.0f\n", m4, m6 );
	}
	return 0;
}