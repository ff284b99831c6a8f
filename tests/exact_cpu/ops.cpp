// TEST INFRASTRUCTURE: the primitives of whisper_amd/csrc/exact_ops.h (the arithmetic of WH_FLAG_PARITY_EXACT) compiled for the CPU
// and exported op by op, so that tests/test_exact_cpu.py can hold them against the reference's own code (oracle/_ref) bit for bit without a GPU.
// Nothing here ships: the product's kernels (whisper_amd/csrc/exact.hip) use the same header on the device. The loops around the primitives
// mirror the loops of the kernels; the model-level graph is driven from Python (tests/exact_model.py).
// Build: g++ -O2 -ffp-contract=off -mf16c -shared -fPIC (tests/test_exact_cpu.py).
#include "../../whisper_amd/csrc/exact_ops.h"
#include <vector>
#include <string.h>
using namespace whx;

extern "C" {

// ggml_init's tables (ggml.c:1375-1385): table[ bits ] for every FP16 bit pattern
void x_tables( uint16_t* gelu, uint16_t* expt )
{
	for( int i = 0; i < 65536; i++ )
	{
		union { h16 h; uint16_t u; } c;
		c.u = (uint16_t)i;
		const float f = (float)c.h;
		const double x = (double)f;
		const float g = (float)( 0.5 * x * ( 1.0 + tanh( 0.79788456080286535587989211986876 * x * ( 1.0 + 0.044715 * x * x ) ) ) );
		c.h = (h16)g;
		gelu[ i ] = c.u;
		c.h = (h16)(float)exp( (double)f );
		expt[ i ] = c.u;
	}
}

// ggml_mul_mat of an FP16 weight [N][K] with FP32 activations [M][K] (ggml.c:4588-4611, :4645-4687): out [M][N]
void x_mul_mat( const uint16_t* W, int N, int K, const float* X, int M, float* out )
{
	std::vector<h16> x16( (size_t)K );
	for( int m = 0; m < M; m++ )
	{
		for( int k = 0; k < K; k++ ) x16[ k ] = toF16( X[ (size_t)m * K + k ] );
		for( int n = 0; n < N; n++ ) out[ (size_t)m * N + n ] = dot16( (const h16*)W + (size_t)n * K, x16.data(), K );
	}
}

void x_norm( const float* x, const float* w, const float* b, float* out, int rows, int n )
{
	for( int r = 0; r < rows; r++ ) normRow( x + (size_t)r * n, w, b, out + (size_t)r * n, n );
}

void x_gelu( const uint16_t* table, const float* x, float* out, int64_t n )
{
	for( int64_t i = 0; i < n; i++ ) out[ i ] = gelu16( (const h16*)table, x[ i ] );
}

// ggml_compute_forward_soft_max_f32 (ggml.c:5026-5096), rows in place
void x_softmax( const uint16_t* expTab, float* p, int rows, int cols )
{
	for( int r = 0; r < rows; r++, p += cols )
	{
		float mx = -INFINITY;
		for( int i = 0; i < cols; i++ ) mx = p[ i ] > mx ? p[ i ] : mx;
		double sum = 0.0;
		for( int i = 0; i < cols; i++ )
		{
			if( p[ i ] == -INFINITY ) p[ i ] = 0.0f;
			else
			{
				const float v = exp16( (const h16*)expTab, p[ i ] - mx );
				sum += (double)v;
				p[ i ] = v;
			}
		}
		const float inv = (float)( 1.0 / sum );
		for( int i = 0; i < cols; i++ ) p[ i ] = p[ i ] * inv;
	}
}

// ggml_compute_forward_flash_attn_f16 for ONE head (ggml.c:5912-6097): q, k [T][64] FP16 (row stride ld halves), v [T][64] FP16 the same way;
// out [T][64] FP32 (row stride ldo floats)
void x_flash_attn( const uint16_t* q_, const uint16_t* k_, const uint16_t* v_, int ld, int T, const uint16_t* expTab, float* out, int ldo )
{
	const h16 *q = (const h16*)q_, *k = (const h16*)k_, *v = (const h16*)v_;
	std::vector<float> S( (size_t)T );
	std::vector<h16> S16( (size_t)T ), vcol( (size_t)T );
	const float scale = (float)( 1.0 / sqrt( 64.0 ) );
	for( int i = 0; i < T; i++ )
	{
		float mx = -INFINITY;
		for( int j = 0; j < T; j++ )
		{
			S[ j ] = dot16( k + (size_t)j * ld, q + (size_t)i * ld, 64 ) * scale;
			mx = S[ j ] > mx ? S[ j ] : mx;
		}
		double sum = 0.0;
		for( int j = 0; j < T; j++ )
		{
			const float e = exp16( (const h16*)expTab, S[ j ] - mx );
			sum += (double)e;
			S[ j ] = e;
		}
		const float inv = (float)( 1.0 / sum );
		for( int j = 0; j < T; j++ ) S16[ j ] = toF16( S[ j ] * inv );
		for( int c = 0; c < 64; c++ )
		{
			for( int j = 0; j < T; j++ ) vcol[ j ] = v[ (size_t)j * ld + c ];
			out[ (size_t)i * ldo + c ] = dot16( vcol.data(), S16.data(), T );
		}
	}
}

// decoder: KQ = mul_mat( K, Q ) for one head (whisper.cpp:1633, ggml.c:4645-4687): K [keys][64] FP16 (row stride ld), Q [N][64] FP32 (row stride ldq)
void x_kq( const uint16_t* k_, int ld, int keys, const float* Q, int ldq, int N, float* S )
{
	h16 q16[ 64 ];
	for( int i = 0; i < N; i++ )
	{
		for( int c = 0; c < 64; c++ ) q16[ c ] = toF16( Q[ (size_t)i * ldq + c ] );
		for( int j = 0; j < keys; j++ ) S[ (size_t)i * keys + j ] = dot16( (const h16*)k_ + (size_t)j * ld, q16, 64 );
	}
}

// decoder: KQV = mul_mat( V_trans, KQ_soft_max ), the transposed-src0 branch (ggml.c:4689-4735) + FINALIZE (:4615-4644), one head:
// P [N][keys] FP32, V [keys][64] FP16 (row stride ld), nth threads; out [N][64] (row stride ldo)
void x_pv_mad( const float* P, const uint16_t* v_, int ld, int keys, int N, int nth, float* out, int ldo )
{
	const h16* v = (const h16*)v_;
	const int dc = ( keys + nth - 1 ) / nth;
	for( int i = 0; i < N; i++ )
		for( int c = 0; c < 64; c++ )
		{
			float total = 0.0f;
			for( int ith = 0; ith < nth; ith++ )
			{
				h16 y = (h16)0.0f;
				const int j1 = dc * ( ith + 1 ) < keys ? dc * ( ith + 1 ) : keys;
				for( int j = dc * ith; j < j1; j++ ) y = mad16( y, v[ (size_t)j * ld + c ], P[ (size_t)i * keys + j ] );
				total = ith == 0 ? toF32( y ) : total + toF32( y );
			}
			out[ (size_t)i * ldo + c ] = total;
		}
}

// ggml_conv_1d_1s / _2s (ggml.c:5199-5318, :5465-5584): W [oc][ic][3] FP16 (the file's layout), X [ic][Tin] FP32, out [oc][Tin / stride]
void x_conv( const uint16_t* W_, int oc, int ic, const float* X, int Tin, int stride, float* out )
{
	const h16* W = (const h16*)W_;
	const int ew0 = ( ic + 31 ) & ~31;
	std::vector<h16> wk( (size_t)3 * ew0 ), xs( (size_t)( Tin + 2 ) * ew0, (h16)0.0f );
	for( int c = 0; c < ic; c++ )
		for( int t = 0; t < Tin; t++ ) xs[ (size_t)( t + 1 ) * ew0 + c ] = toF16( X[ (size_t)c * Tin + t ] );
	const int Tout = Tin / stride;
	for( int o = 0; o < oc; o++ )
	{
		for( size_t i = 0; i < wk.size(); i++ ) wk[ i ] = (h16)0.0f;
		for( int c = 0; c < ic; c++ )
			for( int k = 0; k < 3; k++ ) wk[ (size_t)k * ew0 + c ] = W[ ( (size_t)o * ic + c ) * 3 + k ];
		for( int t = 0; t < Tout; t++ )
		{
			float acc = 0.0f;
			for( int k = 0; k < 3; k++ ) acc += dot16( wk.data() + (size_t)k * ew0, xs.data() + (size_t)( t * stride + k ) * ew0, ew0 );
			out[ (size_t)o * Tout + t ] = acc;
		}
	}
}

}	// extern "C"
