// Does v_cvt_pk_f16_f32 (gfx950) round like v_cvt_f16_f32? All FP32 inputs whose dropped 13 bits are 1000000000000 (ties) plus a random sample.
// hipcc --offload-arch=gfx950 -O2 tools/cvt_probe.hip -o whisper_amd/lib/cvt-probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16;
typedef __attribute__( ( ext_vector_type( 2 ) ) ) _Float16 f16x2;
__global__ void k( const float* x, int n, unsigned* diff, unsigned* first )
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if( i >= n ) return;
	const float a = x[ i ];
	unsigned short s;
	{
		f16 h;
		asm volatile( "v_cvt_f16_f32 %0, %1" : "=v"( h ) : "v"( a ) );
		s = __builtin_bit_cast( unsigned short, h );
	}
	unsigned pk;
	asm volatile( "v_cvt_pk_f16_f32 %0, %1, %1" : "=v"( pk ) : "v"( a ) );
	if( ( pk & 0xffffu ) != s || ( pk >> 16 ) != s )
	{
		const unsigned c = atomicAdd( diff, 1u );
		if( c < 8 )
		{
			first[ 3 * c ] = __builtin_bit_cast( unsigned, a );
			first[ 3 * c + 1 ] = s;
			first[ 3 * c + 2 ] = pk;
		}
	}
}
int main()
{
	const int n = 1 << 24;
	float* h = (float*)malloc( n * 4 );
	uint32_t seed = 12345;
	for( int i = 0; i < n; i++ )
	{
		seed = seed * 1664525u + 1013904223u;
		uint32_t bits = ( seed & 0x0fffe000u ) | 0x38000000u | ( ( seed >> 3 ) & 0x80000000u );	  // exponents around 2^-15 .. 2^16, mantissa high bits random
		if( i & 1 ) bits |= 0x1000u;	  // tie: dropped bits = 1 000000000000
		else bits |= ( seed >> 19 ) & 0x1fffu;
		h[ i ] = __builtin_bit_cast( float, bits );
	}
	float* d; unsigned *diff, *first;
	hipMalloc( &d, n * 4 ); hipMalloc( &diff, 4 ); hipMalloc( &first, 96 );
	hipMemcpy( d, h, n * 4, hipMemcpyHostToDevice ); hipMemset( diff, 0, 4 ); hipMemset( first, 0, 96 );
	hipLaunchKernelGGL( k, dim3( n / 256 ), dim3( 256 ), 0, 0, d, n, diff, first );
	unsigned nd = 0, f[ 24 ];
	hipMemcpy( &nd, diff, 4, hipMemcpyDeviceToHost ); hipMemcpy( f, first, 96, hipMemcpyDeviceToHost );
	printf( "v_cvt_pk_f16_f32 vs v_cvt_f16_f32: %u of %d inputs differ (half of the inputs are ties)\n", nd, n );
	for( unsigned i = 0; i < ( nd < 8 ? nd : 8 ); i++ ) printf( "  x = 0x%08x  cvt = 0x%04x  cvt_pk = 0x%08x\n", f[ 3 * i ], f[ 3 * i + 1 ], f[ 3 * i + 2 ] );
	return 0;
}
