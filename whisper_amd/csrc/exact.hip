// WH_FLAG_PARITY_EXACT: the encoder and decoder graphs with the reference CPU path's arithmetic in the reference's summation order.
//
// Every kernel here is built from the primitives of exact_ops.h (which tests/test_exact_cpu.py holds against oracle/_ref on the CPU): the
// 32-chain dot product of ggml_vec_dot_f16 with its reduction tree, the FP16 multiply-accumulate of ggml_vec_mad_f16 per emulated thread
// range, LayerNorm with sequential double sums, the table softmax. The result is not "close to" the reference: cross-attention caches, logits
// and probabilities are the reference's bits (tests/test_gpu_exact.py), at any thread count -- the thread count only matters to the decoder's
// P.V product (Whisper/source/ggml.c:4689-4735), and it is an argument here.
//
// NEVER timed: one thread per output element and FP32 FMAs on the VALU (no MFMA: the matrix cores sum in their own order). This is what the
// timed kernels (gemm.hip, attn_enc.hip, attn_dec.hip, decode1.hip) are measured against on the device, at any shape and batch size, without
// the reference's own thread-count band in the way. Compiled with -ffp-contract=off: a fused multiply-add happens exactly where fmaf / fma is written.
#include "whisper_hip.h"
#include "kernels.h"
#include "exact_ops.h"

namespace wh
{
	using namespace whx;

	namespace
	{
		// ---- ggml_mul_mat, FP16 weight x FP32 activations (ggml.c:4588-4611, :4645-4687), then the element-wise ops that follow it in
		// whisper.cpp as separate roundings: + bias (ggml_add), * scale (ggml_scale), GELU table (ggml_gelu), + residual (ggml_add).
		// out[ m ][ n ] = epi( dot16( W[ n ], fp16( X[ m ] ) ) ); 16 x 16 outputs per workgroup, one per thread, 32-column K steps through LDS.
		constexpr int XT = 16, XLD = 40;	   // tile edge; LDS row stride in halves (80 bytes: 16-byte aligned, off the bank period)
		__global__ void __launch_bounds__( 256 ) exMulMat( const f16* __restrict__ W, int N, int K, const float* __restrict__ X, long long ldx, int M,
			float* __restrict__ out, long long ldo, const float* __restrict__ bias, float scale, int useScale, const f16* __restrict__ geluTab,
			const float* __restrict__ residual, long long ldr, int altOrder )
		{
			__shared__ __attribute__( ( aligned( 16 ) ) ) f16 Ws[ XT ][ XLD ];
			__shared__ __attribute__( ( aligned( 16 ) ) ) f16 Xs[ XT ][ XLD ];
			const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
			const int n0 = blockIdx.x * XT, m0 = blockIdx.y * XT;
			const int lr = tid >> 4, lc = ( tid & 15 ) * 2;	   // this thread stages row lr, columns lc, lc + 1 of both tiles
			const int wn = min( n0 + lr, N - 1 ), xm = min( m0 + lr, M - 1 );
			const f16* wp = W + (long long)wn * K + lc;
			const float* xp = X + (long long)xm * ldx + lc;
			Dot16 acc;
			acc.clear();
			for( int k0 = 0; k0 < K; k0 += 32 )
			{
				const f16x2 wv = *(const f16x2*)( wp + k0 );
				const float x0 = xp[ k0 ], x1 = xp[ k0 + 1 ];
				__syncthreads();
				*(f16x2*)&Ws[ lr ][ lc ] = wv;
				Xs[ lr ][ lc ] = (f16)x0;
				Xs[ lr ][ lc + 1 ] = (f16)x1;
				__syncthreads();
				acc.step( (const h16*)&Ws[ tx ][ 0 ], (const h16*)&Xs[ ty ][ 0 ] );
			}
			const int n = n0 + tx, m = m0 + ty;
			if( n >= N || m >= M ) return;
			float v = altOrder ? reduceLeftToRight( acc ) : acc.reduce();
			if( bias ) v = bias[ n ] + v;
			if( useScale ) v = v * scale;
			if( geluTab ) v = whx::gelu16( (const h16*)geluTab, v );
			if( residual ) v = v + residual[ (long long)m * ldr + n ];
			out[ (long long)m * ldo + n ] = v;
		}

		// ---- ggml_norm + w * y + b (ggml.c:4098-4156, whisper.cpp:1190-1199): a thread per row, sequential double sums ----
		__global__ void exNorm( const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out, int rows, int n )
		{
			const int r = blockIdx.x * blockDim.x + threadIdx.x;
			if( r >= rows ) return;
			normRow( x + (long long)r * n, w, b, out + (long long)r * n, n );
		}

		// ---- ggml_conv_1d_1s / _2s with an FP16 kernel (ggml.c:5199-5318, :5465-5584): per tap one ggml_vec_dot_f16 over the channels (padded to a
		// multiple of 32 with zeros, which change no chain), the taps added in FP32 in order; then + bias, GELU table.
		// Weights in the arena's layout [oc][ tap * ic + c ] (row stride kpad). Input FP16 [b][ t + 1 ][ c ] with zero rows 0 and Tin + 1 (conv1: the
		// product's own conv input, already fp16( mel )) or FP32 [b][t][c] (conv2: conv1's output, FP16-valued). A thread per output.
		template<bool IN16>
		__global__ void exConv( const f16* __restrict__ W, int kpad, int ic, const void* __restrict__ Xv, long long xBatchStride, int Tin, int stride,
			const float* __restrict__ bias, const f16* __restrict__ geluTab, const float* __restrict__ pe, float* __restrict__ out, long long outBatchStride, int oc )
		{
			const int o = blockIdx.x * blockDim.x + threadIdx.x;
			const int t = blockIdx.y, b = blockIdx.z;
			if( o >= oc ) return;
			float total = 0.0f;
			for( int k = 0; k < 3; k++ )
			{
				const int ti = t * stride + k - 1;
				const bool inside = ti >= 0 && ti < Tin;
				Dot16 acc;
				acc.clear();
				const f16* wr = W + (long long)o * kpad + (long long)k * ic;
				for( int c0 = 0; c0 < ic; c0 += 32 )
				{
#pragma unroll
					for( int i = 0; i < 32; i++ )
					{
						const int c = c0 + i;
						float xv = 0.0f, wv = 0.0f;
						if( c < ic )
						{
							wv = (float)wr[ c ];
							if( inside )
							{
								if constexpr( IN16 ) xv = (float)( (const f16*)Xv )[ b * xBatchStride + (long long)( ti + 1 ) * ic + c ];
								else xv = (float)(f16)( (const float*)Xv )[ b * xBatchStride + (long long)ti * ic + c ];
							}
						}
						acc.step1( i, wv, xv );
					}
				}
				total = total + acc.reduce();
			}
			float v = bias[ o ] + total;
			v = whx::gelu16( (const h16*)geluTab, v );
			if( pe ) v = pe[ (long long)t * oc + o ] + v;	   // conv2: cur = e_pe + transpose( cur ) (whisper.cpp:1167)
			out[ b * outBatchStride + (long long)t * oc + o ] = v;
		}

		// ---- ggml_flash_attn_f16, unmasked (ggml.c:5912-6097): 8 query rows of one (window, head) per workgroup ----
		// q, k, v: FP32 [b * T + t][ d ] (the projections with their biases; rounded to FP16 here like the ggml_cpy into F16 tensors,
		// whisper.cpp:1242-1264). out: FP32 [b * T + t][ d ].
		constexpr int FQ = 8;
		__global__ void __launch_bounds__( 256 ) exFlashAttn( const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ out,
			int T, int d, const f16* __restrict__ expTab )
		{
			extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smem[];
			float* S = (float*)smem;							// [FQ][T]
			f16* P16 = (f16*)( S + FQ * T );					// [FQ][T]
			f16* Q16 = P16 + FQ * T;							// [FQ][64]
			__shared__ double redD[ 256 ];
			__shared__ float redF[ 256 ];
			const int tid = threadIdx.x;
			const int q0 = blockIdx.x * FQ, h = blockIdx.y, b = blockIdx.z;
			const long long base = (long long)b * T * d + h * 64;
			for( int i = tid; i < FQ * 64; i += 256 )
			{
				const int qi = min( q0 + ( i >> 6 ), T - 1 );
				Q16[ i ] = (f16)q[ base + (long long)qi * d + ( i & 63 ) ];
			}
			__syncthreads();
			// scores: a thread per key, the key row held in registers for the 8 queries
			for( int j = tid; j < T; j += 256 )
			{
				h16 kr[ 64 ];
#pragma unroll
				for( int c = 0; c < 64; c++ ) kr[ c ] = (h16)k[ base + (long long)j * d + c ];
#pragma unroll 1
				for( int qi = 0; qi < FQ; qi++ )
				{
					Dot16 acc;
					acc.clear();
					acc.step( kr, (const h16*)Q16 + qi * 64 );
					acc.step( kr + 32, (const h16*)Q16 + qi * 64 + 32 );
					S[ qi * T + j ] = acc.reduce() * 0.125f;	   // ggml_vec_scale_f32 with scale = 1 / sqrt( 64 )
				}
			}
			__syncthreads();
			// softmax of each row: max and the double sum are exact in any order (exact_ops.h)
			for( int qi = 0; qi < FQ; qi++ )
			{
				float mx = -INFINITY;
				for( int j = tid; j < T; j += 256 ) mx = fmaxf( mx, S[ qi * T + j ] );
				redF[ tid ] = mx;
				__syncthreads();
				for( int s = 128; s > 0; s >>= 1 )
				{
					if( tid < s ) redF[ tid ] = fmaxf( redF[ tid ], redF[ tid + s ] );
					__syncthreads();
				}
				mx = redF[ 0 ];
				double sum = 0.0;
				for( int j = tid; j < T; j += 256 )
				{
					const float e = whx::exp16( (const h16*)expTab, S[ qi * T + j ] - mx );
					S[ qi * T + j ] = e;
					sum += (double)e;
				}
				redD[ tid ] = sum;
				__syncthreads();
				for( int s = 128; s > 0; s >>= 1 )
				{
					if( tid < s ) redD[ tid ] += redD[ tid + s ];
					__syncthreads();
				}
				const float inv = (float)( 1.0 / redD[ 0 ] );
				__syncthreads();
				for( int j = tid; j < T; j += 256 ) P16[ qi * T + j ] = toF16( S[ qi * T + j ] * inv );
			}
			__syncthreads();
			// O = V . P16 through ggml_vec_dot_f16 over the keys (chains by key index mod 32, leftovers in double)
			for( int o = tid; o < FQ * 64; o += 256 )
			{
				const int qi = o >> 6, c = o & 63;
				if( q0 + qi >= T ) continue;
				const f16* p = P16 + qi * T;
				const float* vc = v + base + c;
				Dot16 acc;
				acc.clear();
				const int np = T & ~31;
				for( int j0 = 0; j0 < np; j0 += 32 )
				{
#pragma unroll
					for( int i = 0; i < 32; i++ ) acc.step1( i, (float)(f16)vc[ (long long)( j0 + i ) * d ], (float)p[ j0 + i ] );
				}
				double sumf = (double)acc.reduce();
				for( int j = np; j < T; j++ ) sumf += (double)( (float)(f16)vc[ (long long)j * d ] * (float)p[ j ] );
				out[ base + (long long)( q0 + qi ) * d + c ] = (float)sumf;
			}
		}

		// ---- FP32 rows -> the FP16 head-major caches: dst[ ( ( b * H + h ) * rowCap + r0 + r ) * 64 + c ] = fp16( src[ ( b * rowsPer + r ) * d + h * 64 + c ] ) ----
		__global__ void exPackHeads( const float* __restrict__ src, f16* __restrict__ dst, int rowsPer, int rowCap, int r0, int H, long long total )
		{
			const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
			if( i >= total ) return;
			const int d = H * 64;
			const int c = (int)( i % d );
			const long long row = i / d;
			const int r = (int)( row % rowsPer );
			const long long b = row / rowsPer;
			dst[ ( ( b * H + ( c >> 6 ) ) * rowCap + r0 + r ) * 64 + ( c & 63 ) ] = (f16)src[ i ];
		}

		// ---- decoder attention (whisper.cpp:1618-1660, :1715-1748) ----
		// KQ = mul_mat( K, Q ): S[ ( ( s * H + h ) * N + i ) * nKeys + j ] = dot16( K[ j ], fp16( Q[ i ] ) ); masked (self-attention): j > nPast + i -> -inf
		__global__ void exDecScores( const float* __restrict__ Q, const f16* __restrict__ Kc, float* __restrict__ S, int N, int nKeys, int H, int rowCap, int hyp,
			int nPast, int masked )
		{
			const int j = blockIdx.x * blockDim.x + threadIdx.x;
			const int i = blockIdx.y % N, h = blockIdx.y / N, s = blockIdx.z;
			if( j >= nKeys ) return;
			const int d = H * 64;
			const float* qr = Q + ( (long long)s * N + i ) * d + h * 64;
			const f16* kr = Kc + ( ( (long long)( s / hyp ) * H + h ) * rowCap + j ) * 64;
			Dot16 acc;
			acc.clear();
#pragma unroll
			for( int c = 0; c < 32; c++ ) acc.step1( c, (float)kr[ c ], (float)(f16)qr[ c ] );
#pragma unroll
			for( int c = 0; c < 32; c++ ) acc.step1( c, (float)kr[ 32 + c ], (float)(f16)qr[ 32 + c ] );
			float v = acc.reduce();
			if( masked && j > nPast + i ) v = -INFINITY;
			S[ ( ( (long long)s * H + h ) * N + i ) * nKeys + j ] = v;
		}

		// ggml_compute_forward_soft_max_f32 (ggml.c:5026-5096), one workgroup per row, in place (also the vocabulary softmax: src -> dst)
		__global__ void __launch_bounds__( 256 ) exSoftMax( const float* src, float* dst, int cols, const f16* __restrict__ expTab )
		{
			__shared__ double redD[ 256 ];
			__shared__ float redF[ 256 ];
			const int tid = threadIdx.x;
			const float* s = src + (long long)blockIdx.x * cols;
			float* p = dst + (long long)blockIdx.x * cols;
			float mx = -INFINITY;
			for( int j = tid; j < cols; j += 256 ) mx = fmaxf( mx, s[ j ] );
			redF[ tid ] = mx;
			__syncthreads();
			for( int k = 128; k > 0; k >>= 1 )
			{
				if( tid < k ) redF[ tid ] = fmaxf( redF[ tid ], redF[ tid + k ] );
				__syncthreads();
			}
			mx = redF[ 0 ];
			double sum = 0.0;
			for( int j = tid; j < cols; j += 256 )
			{
				const float x = s[ j ];
				float e = 0.0f;
				if( x != -INFINITY )
				{
					e = whx::exp16( (const h16*)expTab, x - mx );
					sum += (double)e;
				}
				p[ j ] = e;
			}
			redD[ tid ] = sum;
			__syncthreads();
			for( int k = 128; k > 0; k >>= 1 )
			{
				if( tid < k ) redD[ tid ] += redD[ tid + k ];
				__syncthreads();
			}
			const float inv = (float)( 1.0 / redD[ 0 ] );
			for( int j = tid; j < cols; j += 256 ) p[ j ] = p[ j ] * inv;
		}

		// KQV = mul_mat( V_trans, KQ_soft_max ): the transposed-src0 branch (ggml.c:4689-4735) and its FINALIZE (:4615-4644). Thread `ith` of nth owns the
		// keys [ dc * ith, min( dc * ( ith + 1 ), nKeys ) ), dc = ceil( nKeys / nth ), and accumulates in FP16 key by key; the partials are added in FP32, thread 0 first.
		__global__ void exDecPV( const float* __restrict__ P, const f16* __restrict__ Vc, float* __restrict__ out, int N, int nKeys, int H, int rowCap, int hyp, int nth )
		{
			const int c = threadIdx.x;	   // 64 threads: one per column of the head
			const int i = blockIdx.x % N, h = blockIdx.x / N, s = blockIdx.y;
			const float* p = P + ( ( (long long)s * H + h ) * N + i ) * nKeys;
			const f16* vr = Vc + ( (long long)( s / hyp ) * H + h ) * rowCap * 64 + c;
			if( nth <= 0 )
			{
				// not the reference's arithmetic: the sum the reference's FP16 accumulation approximates, rounded once (products of an FP16 and an FP32 value are exact in double)
				double acc = 0.0;
				for( int j = 0; j < nKeys; j++ ) acc += (double)(float)vr[ (long long)j * 64 ] * (double)p[ j ];
				out[ ( (long long)s * N + i ) * ( H * 64 ) + h * 64 + c ] = (float)acc;
				return;
			}
			const int dc = ( nKeys + nth - 1 ) / nth;
			float total = 0.0f;
			for( int ith = 0; ith < nth; ith++ )
			{
				h16 y = (h16)0.0f;
				const int j1 = min( dc * ( ith + 1 ), nKeys );
				for( int j = dc * ith; j < j1; j++ ) y = mad16( y, (h16)vr[ (long long)j * 64 ], p[ j ] );
				total = ith == 0 ? (float)y : total + (float)y;
			}
			out[ ( (long long)s * N + i ) * ( H * 64 ) + h * 64 + c ] = total;
		}
	}	// namespace

	int launchExactMulMat( const f16* W, int N, int K, const float* X, long long ldx, int M, float* out, long long ldo, const float* bias, float scale, bool useScale,
		const f16* geluTab, const float* residual, long long ldr, hipStream_t stream )
	{
		if( ( K & 31 ) != 0 || N <= 0 || M <= 0 ) { setError( "exact mul_mat: K must be a multiple of 32" ); return WH_E_INVALIDARG; }
		dim3 grid( ( N + XT - 1 ) / XT, ( M + XT - 1 ) / XT );
		if( grid.y > 65535 ) { setError( "exact mul_mat: too many rows" ); return WH_E_INVALIDARG; }
		hipLaunchKernelGGL( exMulMat, grid, dim3( 256 ), 0, stream, W, N, K, X, ldx, M, out, ldo, bias, scale, useScale ? 1 : 0, geluTab, residual, ldr, g_opt.exactAltOrder );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchExactNorm( const float* x, const float* w, const float* b, float* out, int rows, int n, hipStream_t stream )
	{
		hipLaunchKernelGGL( exNorm, dim3( ( rows + 63 ) / 64 ), dim3( 64 ), 0, stream, x, w, b, out, rows, n );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchExactConv( const f16* W, int kpad, int ic, const void* X, bool in16, long long xBatchStride, int Tin, int stride, const float* bias, const f16* geluTab,
		const float* pe, float* out, long long outBatchStride, int oc, int batch, hipStream_t stream )
	{
		const int Tout = Tin / stride;
		dim3 grid( ( oc + 63 ) / 64, Tout, batch );
		if( in16 )
			hipLaunchKernelGGL( exConv<true>, grid, dim3( 64 ), 0, stream, W, kpad, ic, X, xBatchStride, Tin, stride, bias, geluTab, pe, out, outBatchStride, oc );
		else
			hipLaunchKernelGGL( exConv<false>, grid, dim3( 64 ), 0, stream, W, kpad, ic, X, xBatchStride, Tin, stride, bias, geluTab, pe, out, outBatchStride, oc );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchExactFlashAttn( const float* q, const float* k, const float* v, float* out, int batch, int H, int T, const f16* expTab, hipStream_t stream )
	{
		const size_t lds = (size_t)FQ * T * 4 + (size_t)FQ * T * 2 + FQ * 64 * 2;
		static bool attr = false;
		if( !attr )
		{
			WH_HIP( hipFuncSetAttribute( (const void*)exFlashAttn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096 ) );
			attr = true;
		}
		if( lds > 160 * 1024 - 4096 ) { setError( "exact attention: n_audio_ctx too large" ); return WH_E_INVALIDARG; }
		hipLaunchKernelGGL( exFlashAttn, dim3( ( T + FQ - 1 ) / FQ, H, batch ), dim3( 256 ), lds, stream, q, k, v, out, T, H * 64, expTab );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchExactPackHeads( const float* src, f16* dst, int batch, int rowsPer, int rowCap, int r0, int H, hipStream_t stream )
	{
		const long long total = (long long)batch * rowsPer * H * 64;
		hipLaunchKernelGGL( exPackHeads, dim3( (unsigned)( ( total + 255 ) / 256 ) ), dim3( 256 ), 0, stream, src, dst, rowsPer, rowCap, r0, H, total );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchExactDecAttention( const float* Q, const f16* Kc, const f16* Vc, float* scores, float* out, int seqs, int N, int nKeys, int H, int rowCap, int hyp,
		int nPast, bool masked, int nth, const f16* expTab, hipStream_t stream )
	{
		if( H * N > 65535 || seqs > 65535 ) { setError( "exact decoder attention: too many rows" ); return WH_E_INVALIDARG; }
		hipLaunchKernelGGL( exDecScores, dim3( ( nKeys + 63 ) / 64, H * N, seqs ), dim3( 64 ), 0, stream, Q, Kc, scores, N, nKeys, H, rowCap, hyp, nPast, masked ? 1 : 0 );
		hipLaunchKernelGGL( exSoftMax, dim3( (unsigned)( (long long)seqs * H * N ) ), dim3( 256 ), 0, stream, scores, scores, nKeys, expTab );
		hipLaunchKernelGGL( exDecPV, dim3( H * N, seqs ), dim3( 64 ), 0, stream, scores, Vc, out, N, nKeys, H, rowCap, hyp, nth );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchExactSoftMax( const float* src, float* dst, int rows, int cols, const f16* expTab, hipStream_t stream )
	{
		hipLaunchKernelGGL( exSoftMax, dim3( rows ), dim3( 256 ), 0, stream, src, dst, cols, expTab );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	// ggml_init's tables (ggml.c:1375-1385), built on the host with the host's libm exactly as the reference builds its own: 65536 entries each
	void exactBuildTables( uint16_t* gelu, uint16_t* expt )
	{
		for( int i = 0; i < 65536; i++ )
		{
			union { _Float16 h; uint16_t u; } c;
			c.u = (uint16_t)i;
			const float f = (float)c.h;
			const double x = (double)f;
			const float g = (float)( 0.5 * x * ( 1.0 + tanh( 0.79788456080286535587989211986876 * x * ( 1.0 + 0.044715 * x * x ) ) ) );
			c.h = (_Float16)g;
			gelu[ i ] = c.u;
			c.h = (_Float16)(float)exp( (double)f );
			expt[ i ] = c.u;
		}
	}
}	// namespace wh
