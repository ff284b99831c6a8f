// Normalisation, embedding, layout and softmax/sampling kernels (HBM-bound; one wave per row, shuffle reductions).
#include "kernels.h"

namespace wh
{
	namespace
	{
		// ---- LayerNorm + affine -> FP16 -----------------------------------------------------------------------------
		// norm.hlsl / normFixed.hlsl + fmaRepeat1.hlsl of the reference; numerics of ggml_compute_forward_norm_f32
		// (Whisper/source/ggml.c:4098-4156): mean, centred sum of squares, 1/sqrt(var + 1e-5), then w*y + b as two
		// separate FP32 operations (whisper.cpp:1195-1199), then the FP16 rounding the next weight product applies.
		// The reference sums in double; FP32 two-pass with a wavefront shuffle tree differs by ~1e-7 relative.
		constexpr int LN_MAX_CHUNKS = 8;	 // rows up to 2048 columns

		__global__ void __launch_bounds__( 256 ) layerNormKernel( const float* __restrict__ x, const float* __restrict__ w,
			const float* __restrict__ b, f16* __restrict__ out, int rows, int d )
		{
			const int lane = threadIdx.x & 63;
			const int row = blockIdx.x * 4 + ( threadIdx.x >> 6 );
			if( row >= rows ) return;
			f16* const o = out + (long long)row * d;
			// d <= 1280 for every Whisper size: the 5-chunk instance keeps the register count (and the wasted clamped loads) low
			if( d <= 256 * 4 )	  // (the medium shape: no fifth, clamped chunk -- a quarter more load instructions for nothing; same sums)
				layerNormRows<4, 1>( x + (long long)row * d, 0, 1, w, b, d, lane, [ = ]( int, int c, f16x4 v ) { *(f16x4*)( o + c ) = v; } );
			else if( d <= 256 * 5 )
				layerNormRows<5, 1>( x + (long long)row * d, 0, 1, w, b, d, lane, [ = ]( int, int c, f16x4 v ) { *(f16x4*)( o + c ) = v; } );
			else
				layerNormRows<LN_MAX_CHUNKS, 1>( x + (long long)row * d, 0, 1, w, b, d, lane, [ = ]( int, int c, f16x4 v ) { *(f16x4*)( o + c ) = v; } );
		}

		// ---- mel window -> padded FP16 conv input -----------------------------------------------------------------
		// MelInputTensor::create (Whisper/Whisper/MelInputTensor.cpp:8-63) + convolutionPrep1.hlsl: slice
		// [offset, offset + T) of each spectrogram, zero beyond its end, transposed to time-major and rounded to FP16
		// (the convolution rounds its input, ggml.c:5252-5287). Row 0 and row T+1 stay zero: the conv's zero padding.
		// `wins` non-null: every window names its own spectrogram (pointer, length, offset) -- the streams of a batch scheduler are
		// recordings of different lengths; a null pointer is a window of zeros.
		__global__ void __launch_bounds__( 256 ) melToConvInput( const float* __restrict__ mel, long long melStride, long long melLen,
			const int* __restrict__ melOffsets, const MelWindow* __restrict__ wins, f16* __restrict__ x16, long long xBatchStride, int nMels, int T )
		{
			__shared__ float tile[ 64 ][ 65 ];
			const int b = blockIdx.z;
			const int t0 = blockIdx.x * 64;
			const int c0 = blockIdx.y * 64;
			int off = melOffsets ? melOffsets[ b ] : 0;
			const float* src = mel + (long long)b * melStride;
			if( wins )
			{
				const MelWindow w = wins[ b ];
				src = w.mel;
				melLen = w.mel ? w.len : 0;
				off = w.offset;
			}
			const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
			for( int r = ty; r < 64; r += 4 )
			{
				const int c = c0 + r;
				const long long t = (long long)off + t0 + tx;
				float v = 0.0f;
				if( c < nMels && t0 + tx < T && t < melLen )
					v = src[ (long long)c * melLen + t ];
				tile[ r ][ tx ] = v;
			}
			__syncthreads();
			for( int r = ty; r < 64; r += 4 )
			{
				const int t = t0 + r;
				const int c = c0 + tx;
				if( t < T && c < nMels )
					x16[ (long long)b * xBatchStride + (long long)( t + 1 ) * nMels + c ] = (f16)tile[ tx ][ r ];
			}
		}

		// ---- token + position embedding (addRows.hlsl; whisper.cpp:1544-1548) ---------------------------------------
		__global__ void __launch_bounds__( 256 ) embedKernel( const int* __restrict__ tokens, const f16* __restrict__ te,
			const float* __restrict__ pe, float* __restrict__ x, int rows, int nTok, int nPast, const int* __restrict__ nPastDev, int d,
			int nVocab, int nTextCtx )
		{
			const int row = blockIdx.x;
			// ids and positions index two tables: both are clamped to the tables' extents, so that no value in tokensDev or in the
			// device-resident position can turn into an out-of-range read (the host entry points reject such ids before they get here)
			int tok = tokens[ row ];
			tok = tok < 0 ? 0 : ( tok >= nVocab ? nVocab - 1 : tok );
			int pos = ( nPastDev ? nPastDev[ row / nTok ] : nPast ) + row % nTok;
			pos = pos < 0 ? 0 : ( pos >= nTextCtx ? nTextCtx - 1 : pos );
			for( int c = threadIdx.x; c < d; c += 256 )
				x[ (long long)row * d + c ] = (float)te[ (long long)tok * d + c ] + pe[ (long long)pos * d + c ];
		}

		// ---- block reductions ------------------------------------------------------------------------------------
		template<int NW>
		__device__ __forceinline__ float blockMax( float v, float* sh )
		{
			v = waveReduceMax( v );
			const int w = threadIdx.x >> 6;
			if( ( threadIdx.x & 63 ) == 0 ) sh[ w ] = v;
			__syncthreads();
			float r = sh[ 0 ];
#pragma unroll
			for( int i = 1; i < NW; i++ ) r = fmaxf( r, sh[ i ] );
			__syncthreads();
			return r;
		}
		template<int NW>
		__device__ __forceinline__ double blockSumD( double v, double* sh )
		{
			v = waveReduceSumD( v );
			const int w = threadIdx.x >> 6;
			if( ( threadIdx.x & 63 ) == 0 ) sh[ w ] = v;
			__syncthreads();
			double r = sh[ 0 ];
#pragma unroll
			for( int i = 1; i < NW; i++ ) r += sh[ i ];
			__syncthreads();
			return r;
		}

		// ---- table softmax over rows (softMax.hlsl / softMaxLong.hlsl; ggml.c:5030-5090) ---------------------------
		// p = exp16( x - max ) / sum, sum in double like the reference, -inf -> 0. One 1024-thread block per row.
		__global__ void __launch_bounds__( 1024 ) softMaxRows( const float* in, float* out, int cols )
		{
			__shared__ float shf[ 16 ];
			__shared__ double shd[ 16 ];
			const float* x = in + (long long)blockIdx.x * cols;
			float* y = out + (long long)blockIdx.x * cols;
			float m = -INFINITY;
			for( int c = threadIdx.x; c < cols; c += 1024 ) m = fmaxf( m, x[ c ] );
			m = blockMax<16>( m, shf );
			double s = 0.0;
			for( int c = threadIdx.x; c < cols; c += 1024 )
			{
				const float v = x[ c ];
				const float e = ( v == -INFINITY ) ? 0.0f : exp16( v - m );
				y[ c ] = e;
				s += (double)e;
			}
			s = blockSumD<16>( s, shd );
			const float inv = (float)( 1.0 / s );
			for( int c = threadIdx.x; c < cols; c += 1024 ) y[ c ] = y[ c ] * inv;
		}

		// ---- ContextImpl::sampleBest on the device (Whisper/Whisper/ContextImpl.cpp:71-157) -------------------------
		struct ArgMax
		{
			float v;
			int i;
		};
		__device__ __forceinline__ ArgMax better( ArgMax a, ArgMax b )
		{
			// larger value wins; equal values resolve to the lower index (the reference's partial_sort leaves ties unspecified)
			if( b.v > a.v || ( b.v == a.v && b.i < a.i ) ) return b;
			return a;
		}
		__device__ __forceinline__ ArgMax blockArgMax( ArgMax a, ArgMax* sh )
		{
#pragma unroll
			for( int o = 32; o > 0; o >>= 1 )
			{
				ArgMax b;
				b.v = __shfl_xor( a.v, o, 64 );
				b.i = __shfl_xor( a.i, o, 64 );
				a = better( a, b );
			}
			const int w = threadIdx.x >> 6;
			if( ( threadIdx.x & 63 ) == 0 ) sh[ w ] = a;
			__syncthreads();
			ArgMax r = sh[ 0 ];
			for( int i = 1; i < 16; i++ ) r = better( r, sh[ i ] );
			__syncthreads();
			return r;
		}

		__global__ void __launch_bounds__( 1024 ) sampleBestKernel( const float* __restrict__ probs, int nVocab, int tokenBeg,
			int tokenSot, int tokenSolm, int tokenNot, int forceTimestamp, int isInitial, TokenData* __restrict__ out )
		{
			__shared__ ArgMax sha[ 16 ];
			__shared__ double shd[ 16 ];
			const float* p = probs + (long long)blockIdx.x * nVocab;
			const int tsEnd = isInitial ? min( tokenBeg + 101, nVocab ) : nVocab;

			// best text token and the timestamp statistics
			ArgMax tx = { -1.0f, 0x7fffffff }, ts = { -1.0f, 0x7fffffff };
			double sumTs = 0.0;
			for( int c = threadIdx.x; c < nVocab; c += 1024 )
			{
				const float v = p[ c ];
				if( c < tokenBeg )
					tx = better( tx, ArgMax{ v, c } );
				else if( c < tsEnd )
				{
					ts = better( ts, ArgMax{ v, c } );
					sumTs += (double)v;
				}
			}
			tx = blockArgMax( tx, sha );
			ts = blockArgMax( ts, sha );
			sumTs = blockSumD<16>( sumTs, shd );
			const bool onlyTs = ( sumTs > (double)fmaxf( tx.v, -1.0f ) ) || forceTimestamp;

			// top-4 over the surviving tokens, first one that is not sot / solm / not
			const int lo = onlyTs ? tokenBeg : 0;
			int taken[ 4 ];
			ArgMax pick = { -INFINITY, 0 };
			for( int round = 0; round < 4; round++ )
			{
				ArgMax best = { -INFINITY, 0x7fffffff };
				for( int c = lo + threadIdx.x; c < nVocab; c += 1024 )
				{
					bool skip = c >= tsEnd && c >= tokenBeg;	  // masked by the initial-timestamp cap
					for( int k = 0; k < round; k++ ) skip = skip || ( taken[ k ] == c );
					if( !skip ) best = better( best, ArgMax{ p[ c ], c } );
				}
				best = blockArgMax( best, sha );
				taken[ round ] = best.i;
				pick = best;
				const bool special = best.i == tokenSot || best.i == tokenSolm || best.i == tokenNot;
				if( !special ) break;
			}
			if( threadIdx.x == 0 )
			{
				// NaN logits compare false everywhere and would leave the sentinel index: never hand an out-of-range id to the
				// embedding gather of the next step
				if( pick.i < 0 || pick.i >= nVocab ) pick.i = 0;
				TokenData r;
				r.id = pick.i;
				r.tid = ts.v > -1.0f ? ts.i : 0;
				r.p = pick.v;
				r.pt = (float)( (double)ts.v / ( sumTs + 1e-10 ) );
				r.ptsum = (float)sumTs;
				out[ blockIdx.x ] = r;
			}
		}
		// ---- the `width` best continuations of a sequence under sampleBest's own rules (beam search on hypothesis groups) ----
		// Candidate 0 IS ContextImpl::sampleBest's pick (timestamp-vs-text sum rule, initial-timestamp cap, the top tokens that are
		// sot / solm / not skipped); candidates 1 .. width-1 are the next best tokens under the same mask, specials skipped as well.
		// So a beam of width 1 is the greedy decoder. out: [sequences][width] TokenData (tid / pt / ptsum repeated).
		__global__ void __launch_bounds__( 1024 ) beamCandidatesKernel( const float* __restrict__ probs, int nVocab, int tokenBeg,
			int tokenSot, int tokenSolm, int tokenNot, int forceTimestamp, int isInitial, int width, TokenData* __restrict__ out )
		{
			__shared__ ArgMax sha[ 16 ];
			__shared__ double shd[ 16 ];
			const float* p = probs + (long long)blockIdx.x * nVocab;
			const int tsEnd = isInitial ? min( tokenBeg + 101, nVocab ) : nVocab;
			ArgMax tx = { -1.0f, 0x7fffffff }, ts = { -1.0f, 0x7fffffff };
			double sumTs = 0.0;
			for( int c = threadIdx.x; c < nVocab; c += 1024 )
			{
				const float v = p[ c ];
				if( c < tokenBeg )
					tx = better( tx, ArgMax{ v, c } );
				else if( c < tsEnd )
				{
					ts = better( ts, ArgMax{ v, c } );
					sumTs += (double)v;
				}
			}
			tx = blockArgMax( tx, sha );
			ts = blockArgMax( ts, sha );
			sumTs = blockSumD<16>( sumTs, shd );
			const bool onlyTs = ( sumTs > (double)fmaxf( tx.v, -1.0f ) ) || forceTimestamp;
			const int lo = onlyTs ? tokenBeg : 0;
			constexpr int MAXW = 8;
			int taken[ MAXW + 3 ];
			int nTaken = 0, found = 0;
			// at most width + 3 rounds: the three specials may each cost one
			for( int round = 0; round < width + 3 && found < width; round++ )
			{
				ArgMax best = { -INFINITY, 0x7fffffff };
				for( int c = lo + threadIdx.x; c < nVocab; c += 1024 )
				{
					bool skip = c >= tsEnd && c >= tokenBeg;	  // masked by the initial-timestamp cap
					for( int k = 0; k < nTaken; k++ ) skip = skip || ( taken[ k ] == c );
					if( !skip ) best = better( best, ArgMax{ p[ c ], c } );
				}
				best = blockArgMax( best, sha );
				taken[ nTaken++ ] = best.i;
				const bool special = best.i == tokenSot || best.i == tokenSolm || best.i == tokenNot;
				// sampleBest gives up after four rounds and takes the fourth token whatever it is: only candidate 0 can meet that rule
				if( special && !( found == 0 && round == 3 ) ) continue;
				if( threadIdx.x == 0 )
				{
					TokenData r;
					r.id = ( best.i < 0 || best.i >= nVocab ) ? 0 : best.i;
					r.tid = ts.v > -1.0f ? ts.i : 0;
					r.p = best.v;
					r.pt = (float)( (double)ts.v / ( sumTs + 1e-10 ) );
					r.ptsum = (float)sumTs;
					out[ (long long)blockIdx.x * width + found ] = r;
				}
				found++;
			}
			// fewer than `width` tokens under the mask (cannot happen with a real vocabulary): repeat the last one with probability 0
			for( ; found < width; found++ )
				if( threadIdx.x == 0 )
				{
					TokenData r = out[ (long long)blockIdx.x * width + ( found > 0 ? found - 1 : 0 ) ];
					r.p = 0.0f;
					out[ (long long)blockIdx.x * width + found ] = r;
				}
		}

		// softMaxRows with the ROW IN REGISTERS (round 6, option beam_regs): one read and one write of the row instead of three reads and two writes (a workgroup per row:
		// 40 CUs at 40 rows) -- 37.8 -> 18.3 us at 40 x 51865. The same arithmetic in the same order (e = exp16( x - max ), the double sum thread by thread in column
		// order, p = e * float( 1 / sum )): the same bits. cols <= SC_PER * 1024. Built, measured and NOT kept: the candidate search the same way (51 selects per round and
		// thread where beamCandidatesKernel re-reads the row: 57.6 against 26 us, VALU-bound, a third of the row spilled), and both in one launch (124 us: 288 spill
		// instructions under the 128-register cap of a 1024-thread workgroup).
		constexpr int SC_PER = 51;
		__global__ void __launch_bounds__( 1024 ) softMaxRowsReg( const float* __restrict__ in, float* __restrict__ out, int cols )
		{
			__shared__ float shf[ 16 ];
			__shared__ double shd[ 16 ];
			const float* const x = in + (long long)blockIdx.x * cols;
			float* const y = out + (long long)blockIdx.x * cols;
			float v[ SC_PER ];
			float m = -INFINITY;
	#pragma unroll
			for( int i = 0; i < SC_PER; i++ )
			{
				const int c = threadIdx.x + 1024 * i;
				const float t = x[ c < cols ? c : cols - 1 ];	  // (clamped address + select: a branch per element would serialise the loads)
				v[ i ] = c < cols ? t : -INFINITY;
				m = fmaxf( m, v[ i ] );
			}
			m = blockMax<16>( m, shf );
			double s = 0.0;
	#pragma unroll
			for( int i = 0; i < SC_PER; i++ )
			{
				const float e = ( v[ i ] == -INFINITY ) ? 0.0f : exp16( v[ i ] - m );
				v[ i ] = e;
				s += (double)e;
			}
			s = blockSumD<16>( s, shd );
			const float inv = (float)( 1.0 / s );
	#pragma unroll
			for( int i = 0; i < SC_PER; i++ )
			{
				const int c = threadIdx.x + 1024 * i;
				if( c < cols ) y[ c ] = v[ i ] * inv;
			}
		}
		// ---- self-attention cache rows of sequence parents[j] -> sequence j (beam search: hypotheses change lineage) ----
		// Two launches through a scratch copy, so that a permutation (j <- p while p <- q) reads only rows nobody has overwritten:
		// phase 0: scratch[ j ] = cache[ parents[ j ] ], phase 1: cache[ j ] = scratch[ j ]; sequences with parents[ j ] == j are skipped.
		// grid (heads, sequences, layers x 2 (K, V)); rows x 128 bytes per block.
		__global__ void __launch_bounds__( 256 ) reorderCacheKernel( f16* __restrict__ cacheK, f16* __restrict__ cacheV, f16* __restrict__ scratchK,
			f16* __restrict__ scratchV, const int* __restrict__ parents, int heads, int seqStrideSeqs, int keyStride, int rows, int phase,
			const int* __restrict__ rowsDev )
		{
			const int h = blockIdx.x, j = blockIdx.y, l = blockIdx.z >> 1, kv = blockIdx.z & 1;
			const int p = parents[ j ];
			if( p == j ) return;
			// beam search inside a captured graph: the rows that exist are the sequence's decoder position, in device memory
			if( rowsDev ) rows = min( max( rowsDev[ j ], 0 ), keyStride );
			f16* const cache = kv ? cacheV : cacheK;
			f16* const scratch = kv ? scratchV : scratchK;
			const long long layer = (long long)l * seqStrideSeqs * heads * keyStride * HEAD_DIM;
			const f16* src = ( phase == 0 ? cache + layer + ( (long long)p * heads + h ) * keyStride * HEAD_DIM : scratch + layer + ( (long long)j * heads + h ) * keyStride * HEAD_DIM );
			f16* dst = ( phase == 0 ? scratch : cache ) + layer + ( (long long)j * heads + h ) * keyStride * HEAD_DIM;
			for( int i = threadIdx.x; i < rows * 8; i += 256 ) *(f16x8*)( dst + i * 8 ) = *(const f16x8*)( src + i * 8 );
		}

		// The same move for the hypotheses of ONE WINDOW in ONE launch and without the scratch copy (round 6, option reorder_group): the ranking keeps a hypothesis'
		// parent inside its window's group of `group` consecutive sequences, and row r of sequence j depends on row r of its parent only -- so the thread that owns
		// an element position reads that position from every moved sequence's parent into registers and only then writes the moved sequences: no position is written
		// before its last reader has it. Half the bytes of the two-phase copy (no scratch round trip), one launch instead of two. grid (heads, windows, layers x 2).
		template<int GROUP>
		__global__ void __launch_bounds__( 256 ) reorderCacheGroup( f16* __restrict__ cacheK, f16* __restrict__ cacheV, const int* __restrict__ parents, int heads,
			int seqStrideSeqs, int keyStride, const int* __restrict__ rowsDev )
		{
			const int h = blockIdx.x, w = blockIdx.y, l = blockIdx.z >> 1, kv = blockIdx.z & 1;
			const int base = w * GROUP;
			int par[ GROUP ];
			bool any = false;
	#pragma unroll
			for( int j = 0; j < GROUP; j++ )
			{
				int p = parents[ base + j ] - base;
				p = ( p < 0 || p >= GROUP ) ? j : p;	  // (a parent outside the group cannot come from the ranking kernel; such a slot stays as it is)
				par[ j ] = p;
				any = any || p != j;
			}
			if( !any ) return;
			const int rows = min( max( rowsDev[ base ], 0 ), keyStride );	  // the sequences of a window stand at one position
			f16* const cache = ( kv ? cacheV : cacheK ) + (long long)l * seqStrideSeqs * heads * keyStride * HEAD_DIM;
			auto rowsOf = [ & ]( int j ) -> f16* { return cache + ( (long long)( base + j ) * heads + h ) * keyStride * HEAD_DIM; };
			for( int i = threadIdx.x; i < rows * 8; i += 256 )
			{
				f16x8 val[ GROUP ];
	#pragma unroll
				for( int j = 0; j < GROUP; j++ )
					if( par[ j ] != j ) val[ j ] = *(const f16x8*)( rowsOf( par[ j ] ) + i * 8 );
	#pragma unroll
				for( int j = 0; j < GROUP; j++ )
					if( par[ j ] != j ) *(f16x8*)( rowsOf( j ) + i * 8 ) = val[ j ];
			}
		}

		// ---- beam search: the ranking of a step on the device (see kernels.h; the host version it restates: ContextImpl::decodeWindowBeam) ----
		constexpr int BEAM_CHUNK_FRAMES = 3000;	   // CHUNK_FRAMES of host/hostLoop.h
		// WindowScan::feed (host/hostLoop.h = ContextImpl.cpp:597-673) on the state of one hypothesis; true = its window is over
		__device__ __forceinline__ bool beamFeed( BeamHyp& h, const BeamRules& r, int id )
		{
			if( r.forced ) { h.nTok++; h.i++; return false; }
			if( h.over ) return true;
			if( id > r.tokenBeg )
			{
				// a timestamp token moves the sliding window; going back in time ends the window (the token is not kept)
				const int seekDeltaNew = 2 * ( id - r.tokenBeg );
				if( h.hasTs && h.seekDelta > seekDeltaNew && h.resultLen < h.i ) { h.over = 1; return true; }
				h.seekDelta = seekDeltaNew;
				h.resultLen = h.i + 1;
				h.hasTs = 1;
			}
			h.nTok++;
			const bool endOfAudio = h.hasTs && r.seek + h.seekDelta + 100 >= r.seekEnd;
			if( id == r.tokenEot || ( r.maxTokens > 0 && h.i >= r.maxTokens ) || endOfAudio )
			{
				if( h.resultLen == 0 )
				{
					if( r.seek + h.seekDelta + 100 >= r.seekEnd )
						h.resultLen = h.i + 1;
					else
					{
						h.failed = 1; h.over = 1;
						return true;
					}
				}
				if( r.singleSegment )
				{
					h.resultLen = h.i + 1;
					h.seekDelta = BEAM_CHUNK_FRAMES;
				}
				h.over = 1;
				return true;
			}
			// stuck in a repetition loop: give up on this window
			if( h.i == r.nMax - 1 && ( h.resultLen == 0 || h.seekDelta < BEAM_CHUNK_FRAMES / 2 ) )
			{
				h.failed = 1; h.over = 1;
				return true;
			}
			h.i++;
			if( h.i >= r.nMax ) h.over = 1;
			return h.over != 0;
		}
		__device__ __forceinline__ double beamPerToken( const BeamHyp& h ) { return h.sum / (double)max( 1, h.nTok ); }

		// The ranking of one window by ONE lane: S, cand (the window's slots x width proposals) and logP live in LDS (the kernel below stages them: a lane walking its
		// state in global memory paid a dependent round trip per field, 48 us per step)
		struct BeamProp { int parent, k; double score; };
		// lane 0's working arrays: in LDS as well (private arrays indexed at run time are scratch memory -- a memory round trip per element of the insertion sort)
		struct BeamRankWork
		{
			BeamProp pool[ BEAM_MAX_WIDTH * BEAM_MAX_WIDTH ];
			BeamHyp newLive[ BEAM_MAX_WIDTH ];
			int parentSlot[ BEAM_MAX_WIDTH ], lastTok[ BEAM_MAX_WIDTH ];
		};
		__device__ __forceinline__ void beamRankWindow( BeamWindow& S, BeamRankWork& W, const TokenData* cand, const double* logP, int w, int slots, int width, const BeamRules* __restrict__ rules,
			BeamRecord* __restrict__ records, int maxSteps, int windows, int* __restrict__ parents, int* __restrict__ nextTokens )
		{
			const int base = w * slots;
			if( S.done || S.step >= maxSteps )
			{
				// nothing moves any more: every slot continues itself (the reorder skips it) and feeds its last token again
				for( int b = 0; b < slots; b++ ) parents[ base + b ] = base + b;
				S.done = 1;
				return;
			}
			const BeamRules R = rules[ w ];
			const int step = S.step;
			const bool first = step == 0;
			// ---- the pool: (parent, candidate) with parent score + log p, ranked; ties keep the parent's order, then the candidate's ----
			typedef BeamProp Prop;
			Prop* const pool = W.pool;
			int nPool = 0;
			const int nParents = first ? 1 : S.nLive;	   // the first sample: every slot holds the same prompt, slot 0 speaks for all
			for( int i = 0; i < nParents; i++ )
				for( int k = 0; k < width; k++ )
				{
					const double lp = logP[ i * width + k ];
					pool[ nPool++ ] = Prop{ i, k, ( first ? 0.0 : S.live[ i ].sum ) + lp };
				}
			for( int a = 1; a < nPool; a++ )	   // stable insertion sort, descending score
			{
				const Prop x = pool[ a ];
				int b = a - 1;
				while( b >= 0 && pool[ b ].score < x.score ) { pool[ b + 1 ] = pool[ b ]; b--; }
				pool[ b + 1 ] = x;
			}
			// ---- the best `width` proposals continue their parents: live or, when the stop rules end the window, finished ----
			BeamHyp* const newLive = W.newLive;
			int* const parentSlot = W.parentSlot;
			int* const lastTok = W.lastTok;
			int nNew = 0, accepted = 0;
			BeamRecord* const rec = records + ( (long long)step * windows + w ) * width;
			for( int q = 0; q < nPool && accepted < width; q++ )
			{
				const Prop pr = pool[ q ];
				const TokenData t = cand[ pr.parent * width + pr.k ];
				BeamHyp h;
				if( first )
				{
					h.sum = 0.0; h.i = 0; h.hasTs = 0; h.seekDelta = BEAM_CHUNK_FRAMES; h.resultLen = 0; h.failed = 0; h.over = 0; h.nTok = 0; h.rec = -1;
				}
				else
					h = S.live[ pr.parent ];
				const int parentRec = h.rec;
				h.sum = pr.score;
				const bool over = beamFeed( h, R, t.id );
				h.rec = step * width + accepted;
				rec[ accepted ] = BeamRecord{ t, parentRec, over ? 1 : 0, 0 };
				accepted++;
				if( over )
				{
					if( S.nFinished < BEAM_MAX_FINISHED ) S.finished[ S.nFinished++ ] = h;
				}
				else
				{
					parentSlot[ nNew ] = pr.parent;
					lastTok[ nNew ] = t.id;
					newLive[ nNew++ ] = h;
				}
			}
			for( int a = accepted; a < width; a++ ) rec[ a ] = BeamRecord{ TokenData{ 0, 0, 0.0f, 0.0f, 0.0f }, -2, 0, 0 };	   // unused entries of the step
			for( int i = 0; i < nNew; i++ ) S.live[ i ] = newLive[ i ];
			S.nLive = nNew;
			S.step = step + 1;
			// ---- does a live hypothesis still have a chance? (a cumulative log-probability only ever decreases) ----
			bool keep = nNew > 0;
			if( keep && S.nFinished >= width )
			{
				double bestFinished = -1e300, bestLive = -1e300;
				for( int i = 0; i < S.nFinished; i++ )
					if( !S.finished[ i ].failed ) bestFinished = fmax( bestFinished, S.finished[ i ].sum );
				for( int i = 0; i < nNew; i++ ) bestLive = fmax( bestLive, S.live[ i ].sum );
				keep = bestLive >= bestFinished;
			}
			if( keep && S.nFinished > 2 * width )
			{
				// the finished list keeps its best `width` entries: successful windows first, then log-probability per token (stable)
				for( int a = 1; a < S.nFinished; a++ )
				{
					const BeamHyp x = S.finished[ a ];
					int b = a - 1;
					while( b >= 0 )
					{
						const BeamHyp& y = S.finished[ b ];
						const bool xFirst = x.failed != y.failed ? !x.failed : beamPerToken( x ) > beamPerToken( y );
						if( !xFirst ) break;
						S.finished[ b + 1 ] = y;
						b--;
					}
					S.finished[ b + 1 ] = x;
				}
				S.nFinished = width;
			}
			// the context's end (WindowScan's own bound fires first for every prompt the host loop builds)
			if( keep && S.nPrompt + ( S.step - 1 ) >= S.nTextCtx ) keep = false;
			if( !keep )
			{
				S.done = 1;
				for( int b = 0; b < slots; b++ ) parents[ base + b ] = base + b;
				return;
			}
			// live hypothesis j decodes in slot j next: its cache rows come from its parent's slot; idle slots repeat hypothesis 0 and are ignored
			for( int b = 0; b < slots; b++ )
			{
				const int j = b < nNew ? b : 0;
				parents[ base + b ] = base + parentSlot[ j ];
				nextTokens[ base + b ] = lastTok[ j ];
			}
		}

		// One workgroup per window: 64 lanes stage the window's state, proposals and their log-probabilities (a double-precision log is ~1 us on a single lane) in LDS,
		// lane 0 ranks and runs the state machines, 64 lanes write the state back.
		__global__ void __launch_bounds__( 64 ) beamRankKernel( const TokenData* __restrict__ cand, int slots, int width, const BeamRules* __restrict__ rules,
			BeamWindow* __restrict__ state, BeamRecord* __restrict__ records, int maxSteps, int windows, int* __restrict__ parents, int* __restrict__ nextTokens )
		{
			const int w = blockIdx.x;
			__shared__ BeamWindow S;
			__shared__ TokenData candS[ BEAM_MAX_WIDTH * BEAM_MAX_WIDTH ];
			__shared__ double logP[ BEAM_MAX_WIDTH * BEAM_MAX_WIDTH ];
			__shared__ BeamRankWork work;
			static_assert( sizeof( BeamWindow ) % 4 == 0, "staged as 32-bit words" );
			constexpr int nWords = (int)( sizeof( BeamWindow ) / 4 );
			int* const sw = (int*)&S;
			int* const gw = (int*)&state[ w ];
			for( int i = threadIdx.x; i < nWords; i += 64 ) sw[ i ] = gw[ i ];
			if( (int)threadIdx.x < slots * width )
			{
				const TokenData t = cand[ (long long)w * slots * width + threadIdx.x ];
				candS[ threadIdx.x ] = t;
				logP[ threadIdx.x ] = log( fmax( (double)t.p, 1e-30 ) );
			}
			__syncthreads();
			if( threadIdx.x == 0 ) beamRankWindow( S, work, candS, logP, w, slots, width, rules, records, maxSteps, windows, parents, nextTokens );
			__syncthreads();
			for( int i = threadIdx.x; i < nWords; i += 64 ) gw[ i ] = sw[ i ];
		}

		// ---- logits row -> table softmax -> sampleBest in ONE kernel, the row held in registers -----------------------
		// Used by the captured decode step: no host round trip between the logits product and the next token. Same
		// arithmetic as softMaxRows followed by sampleBestKernel (p = exp16(x - max) * float(1 / double sum)).
		constexpr int SS_PER = 51;	   // ceil( 51866 / 1024 )
		__global__ void __launch_bounds__( 1024 ) softMaxSampleKernel( const float* __restrict__ logits, float* __restrict__ probsOut,
			int nVocab, int tokenBeg, int tokenSot, int tokenSolm, int tokenNot, const DecodeState* __restrict__ state,
			TokenData* __restrict__ out, int* __restrict__ nextTokens, const SampleMailbox mail )
		{
			__shared__ float shf[ 16 ];
			__shared__ double shd[ 16 ];
			__shared__ ArgMax sha[ 16 ];
			const float* x = logits + (long long)blockIdx.x * nVocab;
			const int forceTimestamp = state->forceTimestamp, isInitial = state->isInitial;
			const int tsEnd = isInitial ? min( tokenBeg + 101, nVocab ) : nVocab;
			float v[ SS_PER ];
			float m = -INFINITY;
#pragma unroll
			for( int j = 0; j < SS_PER; j++ )
			{
				const int c = threadIdx.x + j * 1024;
				v[ j ] = c < nVocab ? x[ c ] : -INFINITY;
				m = fmaxf( m, v[ j ] );
			}
			m = blockMax<16>( m, shf );
			double s = 0.0;
#pragma unroll
			for( int j = 0; j < SS_PER; j++ )
			{
				const float e = ( v[ j ] == -INFINITY ) ? 0.0f : exp16( v[ j ] - m );
				v[ j ] = e;
				s += (double)e;
			}
			s = blockSumD<16>( s, shd );
			const float inv = (float)( 1.0 / s );
			ArgMax tx = { -1.0f, 0x7fffffff }, ts = { -1.0f, 0x7fffffff };
			double sumTs = 0.0;
#pragma unroll
			for( int j = 0; j < SS_PER; j++ )
			{
				const int c = threadIdx.x + j * 1024;
				const float p = v[ j ] * inv;
				v[ j ] = p;
				if( c < nVocab )
				{
					if( probsOut ) probsOut[ (long long)blockIdx.x * nVocab + c ] = p;
					if( c < tokenBeg )
						tx = better( tx, ArgMax{ p, c } );
					else if( c < tsEnd )
					{
						ts = better( ts, ArgMax{ p, c } );
						sumTs += (double)p;
					}
				}
			}
			tx = blockArgMax( tx, sha );
			ts = blockArgMax( ts, sha );
			sumTs = blockSumD<16>( sumTs, shd );
			const bool onlyTs = ( sumTs > (double)fmaxf( tx.v, -1.0f ) ) || forceTimestamp;
			const int lo = onlyTs ? tokenBeg : 0;
			ArgMax pick = { -INFINITY, 0 };
			for( int round = 0; round < 4; round++ )
			{
				ArgMax best = { -INFINITY, 0x7fffffff };
#pragma unroll
				for( int j = 0; j < SS_PER; j++ )
				{
					const int c = threadIdx.x + j * 1024;
					const bool ok = c < nVocab && c >= lo && !( c >= tsEnd && c >= tokenBeg );
					if( ok ) best = better( best, ArgMax{ v[ j ], c } );
				}
				best = blockArgMax( best, sha );
				pick = best;
				// remove the winner from its owner's registers for the next round
#pragma unroll
				for( int j = 0; j < SS_PER; j++ )
					if( (int)threadIdx.x + j * 1024 == best.i ) v[ j ] = -INFINITY;
				const bool special = best.i == tokenSot || best.i == tokenSolm || best.i == tokenNot;
				if( !special ) break;
			}
			if( threadIdx.x == 0 )
			{
				// NaN logits compare false everywhere and would leave the sentinel index: never hand an out-of-range id to the
				// embedding gather of the next step
				if( pick.i < 0 || pick.i >= nVocab ) pick.i = 0;
				TokenData r;
				r.id = pick.i;
				r.tid = ts.v > -1.0f ? ts.i : 0;
				r.p = pick.v;
				r.pt = (float)( (double)ts.v / ( sumTs + 1e-10 ) );
				r.ptsum = (float)sumTs;
				const long long slot = (long long)state->step * gridDim.x + blockIdx.x;
				out[ slot ] = r;
				nextTokens[ blockIdx.x ] = pick.i;
				const int gen = state->gen;
				if( mail.data && gen != 0 )
				{
					// host mailbox (pinned): the record as write-through stores, drained, then the stamp the host polls. No release
					// fence: at system scope that is a write-back of the whole L2 (measured +32 us per step at 112 rows); the order
					// data -> stamp is kept by waiting for the data stores before the stamp is issued (MI355X_MICROARCH.md, handoff-flag)
					int* const md = (int*)( mail.data + slot );
					__hip_atomic_store( md + 0, r.id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					__hip_atomic_store( md + 1, r.tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					__hip_atomic_store( md + 2, __float_as_int( r.p ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					__hip_atomic_store( md + 3, __float_as_int( r.pt ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					__hip_atomic_store( md + 4, __float_as_int( r.ptsum ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					// the record validates itself: a word that depends on every field and on the generation travels with it, so a host that
					// ever saw the stamp before all of the record (the order below rests on store acknowledgements, not on a release) reads a
					// checksum that does not match and keeps polling instead of consuming a torn record
					const int check = (int)( (unsigned)gen ^ (unsigned)r.id ^ ( (unsigned)r.tid * 0x9E3779B1u ) ^ (unsigned)__float_as_int( r.p ) ^
						( (unsigned)__float_as_int( r.pt ) * 3u ) ^ ( (unsigned)__float_as_int( r.ptsum ) * 5u ) );
					__hip_atomic_store( mail.flag + 2 * slot + 1, check, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
					__hip_atomic_store( mail.flag + 2 * slot, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
				}
			}
		}

		// ---- the same sampler for the few rows of ONE stream, spread over the chip -------------------------------------
		// softMaxSampleKernel gives a row to one workgroup: 207 KB of logits and 52 k exponentials on ONE CU = 41 us of the ~1.1 ms a
		// single-stream token takes. Here a row is cut into SP_G slices: (1) slice maxima, (2) e = exp16( x - global max ), slice sums,
		// slice argmaxima and top-4 lists, (3) one wave per row merges the SP_G records and applies sampleBest's rules. Same values:
		// p = e * float( 1 / double sum ) for every number that leaves the sampler; the timestamp sum adds the same products in another order
		// (double). e (unnormalised) is left in `eOut`.
		constexpr int SP_G = 64;
		struct SamplePart
		{
			double sumE;
			ArgMax tx, ts;
			ArgMax topAll[ 4 ], topTs[ 4 ];
		};
		__global__ void __launch_bounds__( 256 ) sampleSpreadMax( const float* __restrict__ logits, int nVocab, float* __restrict__ partMax )
		{
			__shared__ float sh[ 4 ];
			const int row = blockIdx.y, g = blockIdx.x;
			const int per = ( nVocab + SP_G - 1 ) / SP_G;
			const int c0 = g * per, c1 = min( c0 + per, nVocab );
			const float* x = logits + (long long)row * nVocab;
			float m = -INFINITY;
			for( int c = c0 + threadIdx.x; c < c1; c += 256 ) m = fmaxf( m, x[ c ] );
			m = waveReduceMax( m );
			if( ( threadIdx.x & 63 ) == 0 ) sh[ threadIdx.x >> 6 ] = m;
			__syncthreads();
			if( threadIdx.x == 0 ) partMax[ row * SP_G + g ] = fmaxf( fmaxf( sh[ 0 ], sh[ 1 ] ), fmaxf( sh[ 2 ], sh[ 3 ] ) );
		}
		__device__ __forceinline__ ArgMax block4ArgMax( ArgMax a, ArgMax* sh )
		{
#pragma unroll
			for( int o = 32; o > 0; o >>= 1 )
			{
				ArgMax b;
				b.v = __shfl_xor( a.v, o, 64 );
				b.i = __shfl_xor( a.i, o, 64 );
				a = better( a, b );
			}
			if( ( threadIdx.x & 63 ) == 0 ) sh[ threadIdx.x >> 6 ] = a;
			__syncthreads();
			const ArgMax r = better( better( sh[ 0 ], sh[ 1 ] ), better( sh[ 2 ], sh[ 3 ] ) );
			__syncthreads();
			return r;
		}
		__global__ void __launch_bounds__( 256 ) sampleSpreadExp( const float* __restrict__ logits, int nVocab, int tokenBeg, const float* __restrict__ partMax,
			const DecodeState* __restrict__ state, float* __restrict__ eOut, SamplePart* __restrict__ parts )
		{
			__shared__ ArgMax sha[ 4 ];
			__shared__ double shd[ 4 ];
			const int row = blockIdx.y, g = blockIdx.x;
			const int per = ( nVocab + SP_G - 1 ) / SP_G;
			const int c0 = g * per, c1 = min( c0 + per, nVocab );
			const float* x = logits + (long long)row * nVocab;
			const int tsEnd = state->isInitial ? min( tokenBeg + 101, nVocab ) : nVocab;
			float m = partMax[ row * SP_G + ( threadIdx.x & ( SP_G - 1 ) ) ];
			m = waveReduceMax( m );
			constexpr int PER_T = 4;	   // ceil( 51866 / 64 / 256 )
			float v[ PER_T ];
			double s = 0.0;
			ArgMax tx = { -1.0f, 0x7fffffff }, ts = { -1.0f, 0x7fffffff };
#pragma unroll
			for( int j = 0; j < PER_T; j++ )
			{
				const int c = c0 + threadIdx.x + j * 256;
				float e = -1.0f;	   // below every real e: never wins an argmax
				if( c < c1 )
				{
					const float xv = x[ c ];
					e = ( xv == -INFINITY ) ? 0.0f : exp16( xv - m );
					eOut[ (long long)row * nVocab + c ] = e;
					s += (double)e;
					if( c < tokenBeg ) tx = better( tx, ArgMax{ e, c } );
					else if( c < tsEnd ) ts = better( ts, ArgMax{ e, c } );
				}
				v[ j ] = e;
			}
			s = waveReduceSumD( s );
			if( ( threadIdx.x & 63 ) == 0 ) shd[ threadIdx.x >> 6 ] = s;
			tx = block4ArgMax( tx, sha );	   // (its barriers also publish shd)
			ts = block4ArgMax( ts, sha );
			SamplePart p;
			p.sumE = ( shd[ 0 ] + shd[ 1 ] ) + ( shd[ 2 ] + shd[ 3 ] );
			p.tx = tx; p.ts = ts;
			// the slice's four best tokens among all that may be sampled, and among its timestamps: whichever list the row needs is merged later
			for( int list = 0; list < 2; list++ )
			{
				float w[ PER_T ];
#pragma unroll
				for( int j = 0; j < PER_T; j++ )
				{
					const int c = c0 + threadIdx.x + j * 256;
					const bool ok = c < c1 && !( c >= tsEnd && c >= tokenBeg ) && ( list == 0 || c >= tokenBeg );
					w[ j ] = ok ? v[ j ] : -2.0f;
				}
				for( int round = 0; round < 4; round++ )
				{
					ArgMax best = { -2.0f, 0x7fffffff };
#pragma unroll
					for( int j = 0; j < PER_T; j++ ) best = better( best, ArgMax{ w[ j ], c0 + (int)threadIdx.x + j * 256 } );
					best = block4ArgMax( best, sha );
					if( best.v < 0.0f ) best.i = 0x7fffffff;	   // the slice has fewer than `round + 1` such tokens
					( list == 0 ? p.topAll : p.topTs )[ round ] = best;
#pragma unroll
					for( int j = 0; j < PER_T; j++ )
						if( c0 + (int)threadIdx.x + j * 256 == best.i ) w[ j ] = -2.0f;
				}
			}
			if( threadIdx.x == 0 ) parts[ row * SP_G + g ] = p;
		}
		__global__ void __launch_bounds__( 64 ) sampleSpreadFinal( const SamplePart* __restrict__ parts, const float* __restrict__ e, int nVocab, int tokenBeg,
			int tokenSot, int tokenSolm, int tokenNot, const DecodeState* __restrict__ state, TokenData* __restrict__ out, int* __restrict__ nextTokens,
			const SampleMailbox mail )
		{
			static_assert( SP_G == 64, "one lane per slice" );
			const int row = blockIdx.x, lane = threadIdx.x;
			const SamplePart p = parts[ row * SP_G + lane ];
			const int forceTimestamp = state->forceTimestamp;
			const int tsEnd = state->isInitial ? min( tokenBeg + 101, nVocab ) : nVocab;
			const double sum = waveReduceSumD( p.sumE );
			const float inv = (float)( 1.0 / sum );
			auto waveBest = [ & ]( ArgMax a ) -> ArgMax
			{
#pragma unroll
				for( int o = 32; o > 0; o >>= 1 )
				{
					ArgMax b;
					b.v = __shfl_xor( a.v, o, 64 );
					b.i = __shfl_xor( a.i, o, 64 );
					a = better( a, b );
				}
				return a;
			};
			ArgMax tx = waveBest( p.tx ), ts = waveBest( p.ts );
			tx.v = tx.v < 0.0f ? -1.0f : tx.v * inv;
			ts.v = ts.v < 0.0f ? -1.0f : ts.v * inv;
			// the timestamp probabilities once more, as the products the one-workgroup sampler adds
			double sumTs = 0.0;
			for( int c = tokenBeg + lane; c < tsEnd; c += 64 ) sumTs += (double)( e[ (long long)row * nVocab + c ] * inv );
			sumTs = waveReduceSumD( sumTs );
			const bool onlyTs = ( sumTs > (double)fmaxf( tx.v, -1.0f ) ) || forceTimestamp;
			// merge the slices' lists: every round the best head wins and its lane moves on
			ArgMax mine[ 4 ];
#pragma unroll
			for( int k = 0; k < 4; k++ ) mine[ k ] = onlyTs ? p.topTs[ k ] : p.topAll[ k ];
			int cursor = 0;
			ArgMax pick = { -INFINITY, 0 };
			for( int round = 0; round < 4; round++ )
			{
				ArgMax head = { -2.0f, 0x7fffffff };
#pragma unroll
				for( int k = 0; k < 4; k++ )
					if( k == cursor ) head = mine[ k ];
				if( head.i == 0x7fffffff ) head.v = -2.0f;
				const ArgMax best = waveBest( head );
				if( best.i == head.i && head.i != 0x7fffffff ) cursor++;
				pick = best;
				const bool special = best.i == tokenSot || best.i == tokenSolm || best.i == tokenNot;
				if( !special ) break;
			}
			if( lane == 0 )
			{
				if( pick.i < 0 || pick.i >= nVocab ) { pick.i = 0; pick.v = 0.0f; }
				TokenData r;
				r.id = pick.i;
				r.tid = ts.v > -1.0f ? ts.i : 0;
				r.p = pick.v * inv;
				r.pt = (float)( (double)ts.v / ( sumTs + 1e-10 ) );
				r.ptsum = (float)sumTs;
				const long long slot = (long long)state->step * gridDim.x + blockIdx.x;
				out[ slot ] = r;
				nextTokens[ blockIdx.x ] = pick.i;
				const int gen = state->gen;
				if( mail.data && gen != 0 )
				{
					int* const md = (int*)( mail.data + slot );
					__hip_atomic_store( md + 0, r.id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					__hip_atomic_store( md + 1, r.tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					__hip_atomic_store( md + 2, __float_as_int( r.p ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					__hip_atomic_store( md + 3, __float_as_int( r.pt ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					__hip_atomic_store( md + 4, __float_as_int( r.ptsum ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					const int check = (int)( (unsigned)gen ^ (unsigned)r.id ^ ( (unsigned)r.tid * 0x9E3779B1u ) ^ (unsigned)__float_as_int( r.p ) ^
						( (unsigned)__float_as_int( r.pt ) * 3u ) ^ ( (unsigned)__float_as_int( r.ptsum ) * 5u ) );
					__hip_atomic_store( mail.flag + 2 * slot + 1, check, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
					asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
					__hip_atomic_store( mail.flag + 2 * slot, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
				}
			}
		}

		__global__ void advanceStateKernel( DecodeState* state, int* seqPos, int rows )
		{
			const int i = blockIdx.x * blockDim.x + threadIdx.x;
			if( i < rows ) seqPos[ i ] += 1;
			if( i == 0 )
			{
				state->step += 1;
				state->forceTimestamp = 0;
				state->isInitial = 0;
			}
		}

		// ragged prompt step: the row of sequence b's last real token, lastPos[b], -> its row nTok - 1 (f16, d % 8 == 0)
		__global__ void __launch_bounds__( 256 ) gatherLastRowsKernel( f16* __restrict__ xn, const int* __restrict__ lastPos, int nTok, int d )
		{
			const int b = blockIdx.x;
			int last = lastPos[ b ];
			last = last < 0 ? 0 : ( last > nTok - 1 ? nTok - 1 : last );
			if( last == nTok - 1 ) return;
			const f16* const src = xn + ( (long long)b * nTok + last ) * d;
			f16* const dst = xn + ( (long long)b * nTok + nTok - 1 ) * d;
			for( int c = threadIdx.x * 8; c < d; c += 256 * 8 ) *(f16x8*)( dst + c ) = *(const f16x8*)( src + c );
		}
	}	// namespace

	int launchLayerNorm( const float* x, const float* w, const float* b, f16* out, int rows, int d, hipStream_t stream )
	{
		if( ( d & 3 ) != 0 || d > 256 * LN_MAX_CHUNKS || rows <= 0 )
		{
			setError( "layerNorm: d must be a multiple of 4, at most 2048" );
			return -1;
		}
		hipLaunchKernelGGL( layerNormKernel, dim3( ( rows + 3 ) / 4 ), dim3( 256 ), 0, stream, x, w, b, out, rows, d );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchMelToConvInput( const float* mel, long long melStride, long long melLen, const int* melOffsets, const MelWindow* wins, f16* x16,
		long long xBatchStride, int nMels, int T, int batch, hipStream_t stream )
	{
		dim3 grid( ( T + 63 ) / 64, ( nMels + 63 ) / 64, batch );
		hipLaunchKernelGGL( melToConvInput, grid, dim3( 256 ), 0, stream, mel, melStride, melLen, melOffsets, wins, x16, xBatchStride, nMels, T );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchEmbed( const int* tokens, const f16* te, const float* pe, float* x, int rows, int nTok, int nPast, const int* nPastDev, int d,
		int nVocab, int nTextCtx, hipStream_t stream )
	{
		hipLaunchKernelGGL( embedKernel, dim3( rows ), dim3( 256 ), 0, stream, tokens, te, pe, x, rows, nTok, nPast, nPastDev, d, nVocab, nTextCtx );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchSoftMaxRows( float* x, int rows, int cols, hipStream_t stream )
	{
		hipLaunchKernelGGL( softMaxRows, dim3( rows ), dim3( 1024 ), 0, stream, x, x, cols );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchVocabSoftMax( const float* logits, float* probs, int rows, int nVocab, hipStream_t stream )
	{
		// option beam_regs: the row in registers (one read and one write instead of three reads and two writes), the same probabilities
		if( g_opt.beamRegs && nVocab <= SC_PER * 1024 )
			hipLaunchKernelGGL( softMaxRowsReg, dim3( rows ), dim3( 1024 ), 0, stream, logits, probs, nVocab );
		else
			hipLaunchKernelGGL( softMaxRows, dim3( rows ), dim3( 1024 ), 0, stream, logits, probs, nVocab );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchSampleBest( const float* probs, int rows, int nVocab, int tokenBeg, int tokenSot, int tokenSolm, int tokenNot,
		int forceTimestamp, int isInitial, TokenData* out, hipStream_t stream )
	{
		hipLaunchKernelGGL( sampleBestKernel, dim3( rows ), dim3( 1024 ), 0, stream, probs, nVocab, tokenBeg, tokenSot, tokenSolm,
			tokenNot, forceTimestamp, isInitial, out );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchBeamCandidates( const float* probs, int rows, int nVocab, int tokenBeg, int tokenSot, int tokenSolm, int tokenNot,
		int forceTimestamp, int isInitial, int width, TokenData* out, hipStream_t stream )
	{
		if( width < 1 || width > 8 ) { setError( "beamCandidates: width must be 1 .. 8" ); return -1; }
		hipLaunchKernelGGL( beamCandidatesKernel, dim3( rows ), dim3( 1024 ), 0, stream, probs, nVocab, tokenBeg, tokenSot, tokenSolm,
			tokenNot, forceTimestamp, isInitial, width, out );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchReorderCache( f16* cacheK, f16* cacheV, f16* scratchK, f16* scratchV, const int* parents, int layers, int sequences, int maxSeq,
		int heads, int keyStride, int rows, hipStream_t stream )
	{
		for( int phase = 0; phase < 2; phase++ )
		{
			hipLaunchKernelGGL( reorderCacheKernel, dim3( heads, sequences, layers * 2 ), dim3( 256 ), 0, stream, cacheK, cacheV, scratchK, scratchV, parents,
				heads, maxSeq, keyStride, rows, phase, (const int*)nullptr );
			WH_HIP( hipGetLastError() );
		}
		return 0;
	}

	int launchReorderCacheDev( f16* cacheK, f16* cacheV, f16* scratchK, f16* scratchV, const int* parents, const int* rowsDev, int layers, int sequences, int maxSeq,
		int heads, int keyStride, int group, hipStream_t stream )
	{
		// option reorder_group: the hypotheses of a window move inside their group, in one launch through registers
		if( g_opt.reorderGroup && rowsDev && group > 1 && ( sequences % group ) == 0 )
		{
			const dim3 grid( heads, sequences / group, layers * 2 );
			switch( group )
			{
	#define WH_RG( G ) case G: hipLaunchKernelGGL( reorderCacheGroup<G>, grid, dim3( 256 ), 0, stream, cacheK, cacheV, parents, heads, maxSeq, keyStride, rowsDev ); WH_HIP( hipGetLastError() ); return 0;
				WH_RG( 2 ) WH_RG( 3 ) WH_RG( 4 ) WH_RG( 5 ) WH_RG( 6 ) WH_RG( 7 ) WH_RG( 8 )
	#undef WH_RG
			}
		}
		for( int phase = 0; phase < 2; phase++ )
		{
			hipLaunchKernelGGL( reorderCacheKernel, dim3( heads, sequences, layers * 2 ), dim3( 256 ), 0, stream, cacheK, cacheV, scratchK, scratchV, parents,
				heads, maxSeq, keyStride, 0, phase, rowsDev );
			WH_HIP( hipGetLastError() );
		}
		return 0;
	}

	int launchBeamRank( const TokenData* cand, int windows, int slots, int width, const BeamRules* rules, BeamWindow* state, BeamRecord* records, int maxSteps,
		int* parents, int* nextTokens, hipStream_t stream )
	{
		if( width < 1 || width > BEAM_MAX_WIDTH || slots < width || slots > BEAM_MAX_WIDTH || windows <= 0 ) { setError( "beamRank: 1 <= width <= slots <= 8" ); return -1; }
		hipLaunchKernelGGL( beamRankKernel, dim3( windows ), dim3( 64 ), 0, stream, cand, slots, width, rules, state, records, maxSteps, windows, parents, nextTokens );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchSoftMaxSample( const float* logits, float* probsOut, int rows, int nVocab, int tokenBeg, int tokenSot, int tokenSolm,
		int tokenNot, const DecodeState* state, TokenData* out, int* nextTokens, SampleMailbox mail, hipStream_t stream )
	{
		if( nVocab > SS_PER * 1024 )
		{
			setError( "softMaxSample: vocabulary larger than 52224" );
			return -1;
		}
		hipLaunchKernelGGL( softMaxSampleKernel, dim3( rows ), dim3( 1024 ), 0, stream, logits, probsOut, nVocab, tokenBeg, tokenSot, tokenSolm,
			tokenNot, state, out, nextTokens, mail );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	size_t sampleSpreadScratchBytes( int rows ) { return (size_t)rows * SP_G * ( sizeof( SamplePart ) + sizeof( float ) ); }
	int launchSoftMaxSampleSpread( const float* logits, float* eOut, int rows, int nVocab, int tokenBeg, int tokenSot, int tokenSolm, int tokenNot,
		const DecodeState* state, TokenData* out, int* nextTokens, SampleMailbox mail, void* scratch, hipStream_t stream )
	{
		if( nVocab > SP_G * 256 * 4 || rows < 1 || !scratch || !eOut ) { setError( "softMaxSampleSpread: bad argument" ); return -1; }
		SamplePart* const parts = (SamplePart*)scratch;
		float* const partMax = (float*)( parts + (size_t)rows * SP_G );
		hipLaunchKernelGGL( sampleSpreadMax, dim3( SP_G, rows ), dim3( 256 ), 0, stream, logits, nVocab, partMax );
		hipLaunchKernelGGL( sampleSpreadExp, dim3( SP_G, rows ), dim3( 256 ), 0, stream, logits, nVocab, tokenBeg, partMax, state, eOut, parts );
		hipLaunchKernelGGL( sampleSpreadFinal, dim3( rows ), dim3( 64 ), 0, stream, parts, eOut, nVocab, tokenBeg, tokenSot, tokenSolm, tokenNot, state, out,
			nextTokens, mail );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchAdvanceState( DecodeState* state, int* seqPos, int rows, hipStream_t stream )
	{
		hipLaunchKernelGGL( advanceStateKernel, dim3( ( rows + 127 ) / 128 ), dim3( 128 ), 0, stream, state, seqPos, rows );
		WH_HIP( hipGetLastError() );
		return 0;
	}

	int launchGatherLastRows( f16* xn, const int* lastPos, int batch, int nTok, int d, hipStream_t stream )
	{
		if( ( d & 7 ) != 0 || batch <= 0 || nTok <= 0 ) { setError( "gatherLastRows: bad shape" ); return -1; }
		hipLaunchKernelGGL( gatherLastRowsKernel, dim3( batch ), dim3( 256 ), 0, stream, xn, lastPos, nTok, d );
		WH_HIP( hipGetLastError() );
		return 0;
	}
}
