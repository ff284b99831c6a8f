// Output addressing and the per-element epilogues of the matrix products (shared by gemm.hip and decode1.hip).
#pragma once
#include "kernels.h"

namespace wh
{
	namespace
	{
		__device__ __forceinline__ long long rowOffset( int m, int Mb, int ld, long long batchStride )
		{
			if( Mb <= 0 ) return (long long)m * ld;
			const int b = m / Mb;
			const int t = m - b * Mb;
			return (long long)b * batchStride + (long long)t * ld;
		}

		// V of the encoder attention is stored in the operand order of attentionEnc's P.V MFMA (attn_enc.hip): per (b, h),
		// blocks of 16 keys x 32 dims hold lane-major 8-half fragments, so a wave reads 1 KiB contiguous per MFMA operand.
		//   index( key, dd ) = ( ( (key >> 4) * 2 + (dd >> 5) ) * 64 + ((key >> 2) & 1) * 32 + (dd & 31) ) * 8 + ((key >> 3) & 1) * 4 + (key & 3)
		__device__ __forceinline__ long long vFragIndex( int key, int dd )
		{
			return ( ( (long long)( key >> 4 ) * 2 + ( dd >> 5 ) ) * 64 + ( ( key >> 2 ) & 1 ) * 32 + ( dd & 31 ) ) * 8 + ( ( key >> 3 ) & 1 ) * 4 + ( key & 3 );
		}

		// One output element. m = global row, n = global column, v = FP32 accumulator.
		template<int EPI>
		__device__ __forceinline__ void epilogueOne( const GemmArgs& a, int m, int n, float v )
		{
			switch( EPI )
			{
			case EPI_F32:
			{
				if( a.bias ) v += a.bias[ n ];
				const long long o = rowOffset( m, a.Mb, a.ldc, a.cBatchStride ) + n;
				if( a.res ) v += a.res[ o ];
				a.out32[ o ] = v;
				break;
			}
			case EPI_F16_GELU:
			{
				const long long o = rowOffset( m, a.Mb, a.ldc, a.cBatchStride ) + n;
				a.out16[ o ] = gelu16( v + a.bias[ n ] );
				break;
			}
			case EPI_CONV2:
			{
				const int b = m / a.Mb;
				const int t = m - b * a.Mb;
				const float g = (float)gelu16( v + a.bias[ n ] );
				a.out32[ (long long)m * a.ldc + n ] = a.pe[ (long long)t * a.N + n ] + g;
				break;
			}
			case EPI_QKV_ENC:
			{
				const int d = a.H * HEAD_DIM;
				const int sel = n / d;
				const int c = n - sel * d;
				const int h = c >> 6, dd = c & 63;
				const int b = m / a.T;
				const int t = m - b * a.T;
				const float x = v + a.bias[ n ];
				const long long bh = (long long)b * a.H + h;
				if( sel == 0 )
					a.q[ ( bh * a.T + t ) * HEAD_DIM + dd ] = (f16)x;
				else if( sel == 1 )
					a.k[ ( bh * a.T + t ) * HEAD_DIM + dd ] = (f16)x;
				else
					a.v[ bh * HEAD_DIM * a.Tpad + vFragIndex( t, dd ) ] = (f16)x;
				break;
			}
			case EPI_CROSS_KV:
			{
				const int d = a.H * HEAD_DIM;
				const int layer = n / ( 2 * d );
				const int c2 = n - layer * 2 * d;
				const int isV = c2 >= d;
				const int c = isV ? c2 - d : c2;
				const int h = c >> 6, dd = c & 63;
				const int b = m / a.T;
				const int t = m - b * a.T;
				const long long o = ( ( ( (long long)layer * a.B + b ) * a.H + h ) * a.T + t ) * HEAD_DIM + dd;
				if( isV )
					a.v[ o ] = (f16)( v + a.bias[ n ] );
				else
					a.k[ o ] = (f16)( v * a.scale );
				break;
			}
			case EPI_QKV_DEC:
			{
				const int d = a.H * HEAD_DIM;
				const int sel = n / d;
				const int c = n - sel * d;
				if( sel == 0 )
				{
					a.q[ (long long)m * d + c ] = (f16)( ( v + a.bias[ n ] ) * a.scale );
					break;
				}
				const int h = c >> 6, dd = c & 63;
				const int b = m / a.nTok;
				const int pos = ( a.nPastDev ? a.nPastDev[ b ] : a.nPast ) + ( m - b * a.nTok );
				const long long o = ( ( (long long)b * a.H + h ) * a.textCtx + pos ) * HEAD_DIM + dd;
				if( sel == 1 )
					a.k[ o ] = (f16)( v * a.scale );
				else
					a.v[ o ] = (f16)( v + a.bias[ n ] );
				break;
			}
			case EPI_Q_DEC:
				a.q[ (long long)m * a.N + n ] = (f16)( ( v + a.bias[ n ] ) * a.scale );
				break;
			}
		}
	}	// namespace
}	// namespace wh
