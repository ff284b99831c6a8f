// Kernel launch interface shared by the .hip translation units and the model/context code.
#pragma once
#include "common.h"

namespace wh
{
	// Runtime switches for A/B measurements of kernel variants inside one process (tools/ab_bench.py, wh_debug_set_tuning).
	enum eTuning : unsigned
	{
		TUNE_GEMV_ROWS4 = 2,	 // 4 weight rows per workgroup for small-N / large-K gemv (else 16)
		TUNE_GEMM_4WAVE = 4,	 // ... of those, one wave per SIMD: 4 waves x (128 x 128), 256 accumulator registers, a hand-pipelined K loop with one barrier per K tile (gemmTiled4)
		TUNE_GEMM_FAST_EPI = 1,	 // gemmTiled8: interior tiles leave through gemmTiled4's lean epilogue (no bounds checks / divisions per row, residual a unit ahead of the stores)
		TUNE_GEMM_8WAVE = 16,	 // big tiled GEMMs: 8 waves x (128 x 64), four phases per K tile, wave rows one barrier apart, counted vmcnt (gemmTiled8)
		TUNE_GEMM_BIG = 8,		 // 256x256x64 tiles for large tiled GEMMs (else 128x128x32 everywhere)
		TUNE_GEMV_SMALLREG = 32,	 // 8-slot gemv instance when a wave's K slice fits (fewer registers, same loads in flight)
		TUNE_GEMM_GL = 64,		 // tiled GEMM stages its tiles global -> LDS directly (128x128x64, swizzled source)
		TUNE_LN_SEPARATE_BIGM = 128,	 // more than 16 decode rows: LayerNorm as its own launch instead of 192-256 redundant prologues
		TUNE_ATTN_XCD = 256,		 // encoder attention: the query blocks of one (sequence, head) run on one XCD
		TUNE_ATTN_DEC_G = 512,		 // decoder attention: 8 lanes per K row (coalesced) instead of a row per thread
		TUNE_FUSE_CROSS_Q = 1024,	 // decode steps: LayerNorm + cross-attention query projection inside the attention kernel
		TUNE_FUSE_SELF_BLOCK = 4096,	 // decode steps: LayerNorm + per-head QKV + cache append + self-attention in one launch
		TUNE_GEMV_LN_BLOCK = 8192,	 // decode steps, 17..32 rows: block-wide LayerNorm prologue in the MLP up-projection instead of a launch
									 // (measured round 2, 28 rows: +11 ms per 28-window batch pass against the separate launch -- OFF)
		TUNE_GEMV_K8 = 16384,		 // decode steps: 8 waves split K when K >= 2048 (the MLP down-projection)
		TUNE_SPLIT_STREAMS = 32768,	 // contexts run the encoder on a second, low-priority stream and the decode chain on a high-priority one
		TUNE_ATTN_ENC_F = 65536,		 // encoder attention: scores recomputed per sweep (32 queries x all keys per wave, K/V tiles through LDS) instead of kept in registers
		TUNE_GEMM_WIDE_EPI = 131072,	 // tiled GEMM: accumulators leave through LDS as 16-byte row stores instead of 2/4-byte column stores
		TUNE_GEMM_FRAGPF = 262144,	 // direct-to-LDS tiled GEMM: MFMA fragments of k-substep s+1 are read before the MFMAs of substep s (two register sets, counted LDS waits)
		TUNE_ATTN_ENC_2SWEEP = 524288,	 // attentionEncF: row sum and P.V in one sweep with the unnormalised FP16 e, O scaled by 1 / sum at the end
		TUNE_MEL_MFMA = 8388608,		 // spectrogram: DFT and filterbank as v_mfma_f64_16x16x4_f64 tiles (16 frames per workgroup) instead of FP64 FMAs
		TUNE_SELF_MFMA = 4194304,		 // selfBlockDec: the head's Q/K/V rows as MFMA tiles instead of 8 lanes per weight row on the VALU
		TUNE_GEMV_ALLROWS = 1048576,	 // 33 .. 128 decode rows, N >= 16384 (vocabulary projection): 32 columns x all rows per workgroup (gemmAllRows)
		TUNE_GEMV_ROWGROUPS = 2097152,	 // 33 .. 128 decode rows: 32 instead of 64 rows per workgroup while that leaves fewer than 256 workgroups
		TUNE_GEMM_GROUP_M = 2048,	 // tiled GEMM: blocks walk bands of 4 M tiles (A band stays in the XCD's L2) instead of rows of tiles
		TUNE_ATTN_DEC_NT = 67108864,	 // decode cross-attention: K / V rows (read once per launch, by one workgroup) with the non-temporal load policy
		TUNE_DECODE_SMALL = 16777216,	 // single-token steps of up to 4 sequences: the chip-wide launches of decode1.hip (gemvSmall, cross-attention over 8 key ranges)
		TUNE_DECODE_PREFETCH = 33554432,	 // ... each carrying 256 workgroups that pull the next launch's weights into the L2 of the XCD that will read them
										 // (measured round 3, medium shape, one sequence: 1216 vs 1129 us per token -- OFF)
		TUNE_ATTN_ENC_TABLE = 268435456,	 // encoder attention: exp16 as a LOOKUP in the reference's own table held in LDS (1024-thread workgroups, 4 VALU slots + one
										 // ds_read_u16 per score instead of 14 VALU slots); bit-exact table semantics
		TUNE_ATTN_ENC_TABLE_ANY = 1073741824,	 // ... whatever the grid (tests: small shapes through the table kernel)
		TUNE_SAMPLE_SPREAD = 2147483648u,	 // up to 4 sequences: the sampler's row cut into 64 slices over the chip (3 launches) instead of one workgroup per row
										 // (measured round 4, one stream through runFull: 417.9 / 419.5 without, 418.5 / 417.1 audio-s/s with -- nothing; OFF)
		TUNE_GEMV_MT8 = 536870912,		 // 65 .. 128 decode rows, N / 16 >= 256: all rows in one workgroup (weights streamed once), 4 fragment slots
		TUNE_ENC_SERIAL = 134217728,	 // several contexts in flight: their ENCODERS run one at a time (an event chain between the contexts' streams), so that a
										 // batch's MFMA-bound encoder runs under the latency- and HBM-bound decode chain of its neighbours instead of next to their encoders
		// Chosen from interleaved in-process runs on one MI355X (tools/ab_bench.py, WH_TUNING=<mask> python bench.py;
		// profiles/r01_ab_variants.txt). Retired after measuring slower, ms per clip pass: 8-wave
		// LayerNorm prologue (+3.4, spills), 4-row workgroups for K = d (+0.5), cross-attention split over 4 workgroups with
		// the combine in the next gemv's prologue (+6.7), all of a head's K/V requested up front (+1.5).
		TUNE_DEFAULT = TUNE_GEMM_FAST_EPI | TUNE_GEMM_8WAVE | TUNE_GEMV_ROWS4 | TUNE_GEMV_SMALLREG | TUNE_GEMM_BIG | TUNE_GEMM_GL | TUNE_LN_SEPARATE_BIGM | TUNE_ATTN_XCD | TUNE_ATTN_DEC_G | TUNE_FUSE_CROSS_Q | TUNE_GEMM_GROUP_M | TUNE_FUSE_SELF_BLOCK | TUNE_GEMV_K8 | TUNE_ATTN_ENC_F | TUNE_GEMM_WIDE_EPI | TUNE_GEMM_FRAGPF | TUNE_ATTN_ENC_2SWEEP | TUNE_GEMV_ALLROWS | TUNE_GEMV_ROWGROUPS | TUNE_SELF_MFMA | TUNE_MEL_MFMA | TUNE_DECODE_SMALL | TUNE_ATTN_DEC_NT | TUNE_ATTN_ENC_TABLE
	};
	extern unsigned g_tuning;

	// Integer knobs beyond the 32 tuning bits: wh_debug_set_option( name, value ) in a process, WH_OPT_<NAME> in the environment at load.
	struct Options
	{
		int decTile = 0;			 // "dec_tile": decode products of 129 .. 512 rows: 0 = tile by shape, 44 / 42 / 24 / 22 = 16 MT rows x 16 CT columns pinned, 1 = gemvFused row groups
		int decDepth = 0;			 // "dec_depth": gemmDecRows: 0 = k-steps in flight by tile shape (2 / 3 / 4), 2 = two everywhere
		int decWideRows = 1;		 // "dec_wide_rows": 33 .. 128 rows x N >= 2048 (MLP up-projection, QKV): all rows in one row tile per 32 columns (gemmDecRows) instead of gemvFused
									 // (0 = gemvFused; 2 = also the FP32 epilogue, for tests). Measured: two contexts of 70 / 112 windows +1.4 % / +1.7 % on the job
		int decDeepRows = 0;		 // "dec_deep_rows": 33 .. 128 rows x N <= 2048, K >= 2048 (MLP down-projection): all rows per 16-column workgroup, 8 waves over K (gemmDecRows<.., NW = 8>)
		int vocabDecRows = 0;		 // "vocab_decrows": more than 128 sequences: the vocabulary product through gemmDecRows instead of the M-tiled kernel
		int encChunk = 128;			 // "enc_chunk": the most windows ONE encoder pass takes; contexts created afterwards encode larger batches in equal chunks
		int selfFuseMaxRows = 32;	 // "self_fuse_max_rows": selfBlockDec up to this many sequences, LayerNorm + QKV product + attention launches beyond
									 // (measured, profiles/r05_ab_variants.txt: 224 windows 68.7 us fused vs 17.5 + 10.1 + 2.3 us, 448 windows 127 vs 45 us; two contexts of
									 // 70 / 112 windows: +1.6 % / +3.6 % on the job with the self-attention as selfAttnDecWave -- round 2 measured the fused launch ahead
									 // when the separate attention was attentionDecG at 43 us)
		int selfNq = 0;				 // "self_nq": sequences per selfBlockDec workgroup (0 = by grid size; 1, 2, 4, 8)
		int exactEncLayers = -1;	 // "exact_enc_layers": WH_FLAG_PARITY_EXACT, debugging: encode only this many layers and stop (buffers readable as "exact:<name>"); -1 = all
		int encExp = 5;				 // "enc_exp": encoder attention on the timed path: 5 = attentionEncW<2, true, true> (one sweep, lazily raised running maximum, e = fp16( 2^( s log2 e - m log2 e ) ) with NO FP16 rounding of the argument: 1.46-1.49 ms per 112 windows isolated), 3 = the same with the reference's fp16( s - m ) argument (1.59-1.65), 2 = its two-sweep form (1.85), 1 = attentionEncT<1> (two sweeps, v_exp_f32; 1.83), 0 = attentionEncT<0> (the reference's table in LDS, bit-exact e; 1.91-2.14: round 5's kernel)
		int encAblate = 0;			 // "enc_ablate": attentionEncW, measurement only (results wrong): 1 = no exponentials, 2 = no P.V MFMAs, 4 = a quarter of the Q.K MFMAs
		int exactAltOrder = 0;		 // "exact_alt_order": WH_FLAG_PARITY_EXACT, measurement only: the weight products add their 32 chains left to right instead of in ggml's tree
		int decLds = 1;				 // "dec_lds": decode products of 129 .. 512 rows: 1 = gemmDecTile where its 64 x 64 / 64 x 32 tiles fill the chip (operands staged through LDS in
									 // full 128-byte lines, the same sums: 448 x 4096 x 1024 14.4 against 22.1 us) and, with 4 / 6 / 8 row tiles per workgroup, for the wide
									 // products of 33 .. 128 rows (128 x 4096 x 1024: 10.2 against 15.7 us); 0 = gemmDecRows everywhere (round 5)
		int decLdsKs = 2;			 // "dec_lds_ks": 2 = two K tiles per ring slot and barrier where that pays -- gemmDecTile on 64 x 32 tiles with K >= 2048 (18.1 against 21.7 us at 448 rows)
									 // and the wide products of 33 .. 96 rows (40 x 5120 x 1280: 7.8 against 8.9 us) --, 1 = one tile per slot everywhere
		int decSplit = 1;			 // "dec_split": the MLP down-projection (N <= 2048, K >= 2048) of 33 .. 128 rows: 1 = the eight K shares of gemvFused's eight waves on eight workgroups of
									 // gemmDecTile per 32 columns + decSplitCombine (the same bits, two launches), 0 = gemvFused<.., 8 waves> (rounds 4-5)
		int crossMfma = 1;			 // "cross_mfma": the cross-attention of a decode step for hypothesis groups (beam search): 1 = attentionDecM (scores and P.V on the matrix cores, every
									 // load of the first half in flight at once), 0 = attentionDecG<NQ, true> (rounds 2-5)
		int vocabLds = 1;			 // "vocab_lds": the vocabulary product of 33 .. 128 rows: 1 = gemmDecTile's 64 x 64 tiles (one or two row tiles), 0 = gemmAllRows (32 columns x all rows per workgroup, rounds 3-5)
		int beamRegs = 1;			 // "beam_regs": the vocabulary softmax over rows (beam steps, wh_op_soft_max, wh_decode's probabilities): 1 = the row in registers (softMaxRowsReg: one
									 // read and one write, the same bits: 37.8 -> 18.3 us at 40 rows), 0 = softMaxRows (three reads, two writes)
		int reorderGroup = 1;		 // "reorder_group": the ranked beam step's cache reorder: 1 = reorderCacheGroup (a window's hypotheses in one launch through registers), 0 = the two-phase
									 // copy through the scratch cache (round 4)
		int gemmBigMinRows = 8192;	 // "gemm_big_min_rows": products of at least this many rows (and 300 tiles of 256 x 256) take the persistent 256 x 256 kernel (rounds 3-5: 16384;
									 // 8 windows = 12000 rows: beam5 1341 -> 1370 audio-s/s with 8192, the same ids)
		int gemmMf16 = 1;			 // "gemm_mf16": 1 = gemmTiled8's K loop on v_mfma_f32_16x16x32_f16 (same bits as the 32x32x16 form, +9 % on the class in the model:
									 // profiles/r06_evidence/gemm_vendor_gap.txt); 0 = v_mfma_f32_32x32x16_f16 (rounds 3-5)
		int selfWaveMinRows = 32;	 // "self_wave_min_rows": single-token causal self-attention as its own launch: a wave per (sequence, head) beyond this many sequences
	};
	extern Options g_opt;

	// ---------------------------------------------------------------------------------------------------------------
	// NT GEMM: acc[m][n] = sum_k A[m][k] * W[n][k], FP16 operands, FP32 accumulate on MFMA (32x32x16).
	// Replaces mulMatTiled.hlsl / mulMatByRowTiled.hlsl and, through the row mapping below, the five convolution*.hlsl
	// shaders. Every elementwise shader of the reference that follows a product (addRepeat*, addRepeatGelu,
	// scaleInPlace, copyConvert, copyTranspose, addInPlace) is an epilogue here.
	// ---------------------------------------------------------------------------------------------------------------
	enum eEpilogue : int
	{
		EPI_F32 = 0,	  // out32 = acc (+bias[n]) (+res)                       mulMat + addRepeat(+Ex)
		EPI_F16_GELU,	  // out16 = gelu16( acc + bias[n] )                     mulMat + addRepeatGelu
		EPI_CONV2,		  // out32 = float( gelu16( acc + bias[n] ) ) + pe[t][n]  conv_1d_2s + gelu + positional embedding
		EPI_QKV_ENC,	  // head-split FP16 Q, K and transposed V               encoder Q/K/V + copyConvert/copyTranspose
		EPI_CROSS_KV,	  // cross-attention caches of all decoder layers        WhisperContext.cpp:345-389
		EPI_QKV_DEC,	  // decoder: scaled Q, append scaled K and V to self-KV WhisperContext.cpp:424-452
		EPI_Q_DEC,		  // decoder cross-attention query: fp16( (acc+bias)*scale )
	};

	struct GemmArgs
	{
		const f16* A;
		const f16* W;
		int M, N, K;
		// A row m lives at A + (m / Mb) * aBatchStride + (m % Mb) * lda   (Mb >= M means one segment)
		int lda, Mb;
		long long aBatchStride;
		int epi;
		const float* bias;	  // [N] or null
		const float* res;	  // same addressing as out32, or null
		float* out32;
		f16* out16;
		// output row m lives at (m / Mb) * cBatchStride + (m % Mb) * ldc
		int ldc;
		long long cBatchStride;
		const float* pe;	  // EPI_CONV2: [Mb][N]
		float scale;
		// head-split outputs
		f16* q;				  // ENC: [b][h][T][64]   DEC: [m][d]
		f16* k;				  // ENC: [b][h][T][64]   CROSS: [layer][b][h][T][64]   DEC: self-K [b][h][textCtx][64] (layer applied)
		f16* v;				  // ENC: [b][h][64][Tpad] CROSS: like k                 DEC: self-V like k
		int T, Tpad, H, B;
		int nTok, nPast, textCtx;
		// decode-step extras (launchGemv only)
		const int* nPastDev;  // when non-null the positions come from device memory (graph replay): nPastDev[sequence], else nPast for all
		const float* lnX;	  // when non-null: A is produced on the fly as fp16( LayerNorm(lnX[m]) * lnW + lnB ), row length K
		const float* lnW;
		const float* lnB;
		int wideEpi;		  // tiled kernel, set by the launcher: the LDS-transposed epilogue with 16-byte stores applies
		int groupM;			  // tiled kernel: M tiles per band of the block walk (0 = default for the tile shape, 1 = rows of tiles)
		float* splitScratch;  // decode-step products, option dec_split: [8][M][N] partial tiles of the K-split launch (the context's; null = the split path is not taken)
		int cuLimit;		  // persistent tiled kernel: CUs the launch stream may use (0 = all of the device)
	};

	int launchGemm( const GemmArgs& a, hipStream_t stream );		// M-tiled kernel, any M
	int launchGemmSkinny( const GemmArgs& a, hipStream_t stream );	// M <= 32 rows, any K % 64 == 0: weights streamed once
	// the decode-step kernels, K % 128 == 0. M <= 128 rows: 16 (or 4) weight rows per workgroup, every load of a wave in flight at once,
	// optional fused LayerNorm prologue (up to 32 rows); 129 .. 512 rows: 64 x 64 output tiles, FP16 activation rows only
	int launchGemv( const GemmArgs& a, hipStream_t stream );
	constexpr int GEMV_FUSED_MAX_ROWS = 128;	 // gemvFused / gemmAllRows: 16 (32) columns x up to 128 rows per workgroup
	constexpr int GEMV_MAX_ROWS = 512;		 // beyond that, up to here: gemmDecRows (64 x 64 output tiles, the lock-step batches of 129 .. 512 sequences)
	int launchGemmVariant( const GemmArgs& a, int variant, hipStream_t stream );	// tile-shape experiments, EPI_F32 only
	int gemmInit();													// one-time function attributes

	// ---------------------------------------------------------------------------------------------------------------
	// elementwise / normalisation
	// ---------------------------------------------------------------------------------------------------------------
	// out16[row] = fp16( norm(x[row]) * w + b ); rows of length d (multiple of 4, <= 2048). norm.hlsl + fmaRepeat1.hlsl
	int launchLayerNorm( const float* x, const float* w, const float* b, f16* out, int rows, int d, hipStream_t stream );
	// x16[b][t+1][c] = fp16( mel[b][c][off_b + t] ), zero outside the spectrogram; rows 0 and T+1 are the conv padding.
	// wins (device, [batch]) non-null: window b comes from its own spectrogram wins[b] = { [nMels][len] FP32, len, offset }
	struct MelWindow
	{
		const float* mel;
		long long len;
		int offset, reserved;
	};
	int launchMelToConvInput( const float* mel, long long melStride, long long melLen, const int* melOffsets, const MelWindow* wins,
		f16* x16, long long xBatchStride, int nMels, int T, int batch, hipStream_t stream );
	// x[m] = float(te[token[m]]) + pe[pos(m / nTok) + m % nTok], pos(b) = nPastDev ? nPastDev[b] : nPast   (addRows.hlsl)
	// token ids are clamped to [0, nVocab), positions to [0, nTextCtx): no input can index outside the two tables
	int launchEmbed( const int* tokens, const f16* te, const float* pe, float* x, int rows, int nTok, int nPast, const int* nPastDev, int d,
		int nVocab, int nTextCtx, hipStream_t stream );
	// in-place table softmax of FP32 rows (softMax*.hlsl with the CPU path's FP16 exp table semantics)
	int launchSoftMaxRows( float* x, int rows, int cols, hipStream_t stream );

	// ---------------------------------------------------------------------------------------------------------------
	// WH_FLAG_PARITY_EXACT (exact.hip): the reference CPU path's arithmetic in the reference's summation order, built from exact_ops.h. Never timed.
	// ---------------------------------------------------------------------------------------------------------------
	// out[m][n] = epi( ggml_vec_dot_f16( W[n], fp16( X[m] ) ) ), epi = [bias +] [* scale] [GELU table] [+ residual], each a rounding of its own
	int launchExactMulMat( const f16* W, int N, int K, const float* X, long long ldx, int M, float* out, long long ldo, const float* bias, float scale, bool useScale,
		const f16* geluTab, const float* residual, long long ldr, hipStream_t stream );
	int launchExactNorm( const float* x, const float* w, const float* b, float* out, int rows, int n, hipStream_t stream );
	// k = 3, padding 1; X: FP16 [b][t+1][ic] with zero rows around (in16) or FP32 [b][t][ic]; out FP32 [b][t][oc] = [pe +] GELU( bias + conv )
	int launchExactConv( const f16* W, int kpad, int ic, const void* X, bool in16, long long xBatchStride, int Tin, int stride, const float* bias, const f16* geluTab,
		const float* pe, float* out, long long outBatchStride, int oc, int batch, hipStream_t stream );
	int launchExactFlashAttn( const float* q, const float* k, const float* v, float* out, int batch, int H, int T, const f16* expTab, hipStream_t stream );
	int launchExactPackHeads( const float* src, f16* dst, int batch, int rowsPer, int rowCap, int r0, int H, hipStream_t stream );
	// scores [seqs][H][N][nKeys] scratch; sequence s reads the cache of s / hyp; nth = the reference's thread count (it partitions the keys of P.V)
	int launchExactDecAttention( const float* Q, const f16* Kc, const f16* Vc, float* scores, float* out, int seqs, int N, int nKeys, int H, int rowCap, int hyp,
		int nPast, bool masked, int nth, const f16* expTab, hipStream_t stream );
	int launchExactSoftMax( const float* src, float* dst, int rows, int cols, const f16* expTab, hipStream_t stream );
	void exactBuildTables( uint16_t* gelu, uint16_t* expt );

	// ---------------------------------------------------------------------------------------------------------------
	// attention
	// ---------------------------------------------------------------------------------------------------------------
	// encoder (unmasked) attention, all keys; q,k [bh][T][64], vT [bh][64][Tpad], out [b][T][H*64] FP16
	// exactP: the P.V operand is the reference's fp16( e / sum ) (ggml.c:6035-6046) even when the two-sweep kernel is selected
	// expTab: the model's table of fp16( expf( -|x| ) ) over the FP16 bit patterns 0 .. EXP_TABLE_ENTRIES-1 (null: the arithmetic exp16 kernels only)
	constexpr int EXP_TABLE_ENTRIES = 0x5000;
	constexpr int EXP_TABLE_LIMIT = 0x4C56;	   // first entry that is 0: fp16( expf( -17.34375 ) ); every larger magnitude maps here
	int launchAttentionEnc( const f16* q, const f16* k, const f16* vT, f16* out, int batch, int heads, int T, int Tpad, bool exactP, const f16* expTab,
		hipStream_t stream );
	int attentionInit();

	struct DecAttnArgs
	{
		const f16* q;		   // [batch*nTok][d], already scaled
		const f16* kc;		   // [batch][H][keyStride][64]
		const f16* vc;
		f16* out;			   // [batch*nTok][d]
		int batch, H, nTok;
		int nKeys;			   // keys visible to the LAST query (self: nPast+nTok; cross: n_audio_ctx)
		int keyStride;		   // rows allocated per (b,h)
		int causal;			   // 1: query i sees keys <= nPast + i
		int nPast;
		int parityThreads;	   // 0 = FP32 P.V; >0 = emulate ggml's FP16 thread-partitioned accumulation
		const int* nPastDev;   // causal only: when non-null nPast (and nKeys = nPast + nTok) of sequence b come from device memory, nPastDev[b]
		// `group` consecutive sequences share one K/V block (the hypotheses of a window in cross-attention): kc / vc hold
		// batch / group blocks and the rows b*group .. b*group + group - 1 are served by one pass over block b. 0 or 1 = none.
		int group;
		// fused query (cross-attention, decode steps): when lnX is non-null, q is ignored and the query of row m is
		// fp16( ( qW[h*64 + j] . fp16( LayerNorm( lnX[m] ) * lnW + lnB ) + qB[h*64 + j] ) * qScale )
		const float* lnX;	   // [batch*nTok][d] residual stream
		const float* lnW;
		const float* lnB;
		const f16* qW;		   // [d][d]
		const float* qB;	   // [d]
		float qScale;
	};
	int launchAttentionDec( const DecAttnArgs& a, hipStream_t stream );

	// The self-attention half of a single-token decode step as one launch: LayerNorm -> per-head Q/K/V rows of the fused
	// [3d][d] weight -> cache append at the current position -> causal attention -> out (FP16 [batch][d]).
	struct DecSelfArgs
	{
		const float* x;		   // [batch][d] residual stream
		const float* lnW;
		const float* lnB;
		const f16* wqkv;	   // [3d][d]: query, key, value rows
		const float* bqkv;	   // [3d] (the key third is zero)
		float scale;		   // (d/H)^-0.25 on q and k
		f16* kc;			   // self-attention caches of this layer [batch][H][keyStride][64]
		f16* vc;
		f16* out;
		int batch, H, keyStride;
		int nPast;
		const int* nPastDev;   // when non-null the position of sequence b comes from device memory (graph replay): nPastDev[b]
	};
	int launchSelfBlockDec( const DecSelfArgs& a, hipStream_t stream );


	// ---------------------------------------------------------------------------------------------------------------
	// decode steps of ONE stream (up to 4 sequences), decode1.hip: every launch covers the chip, and every launch pulls the
	// weights of the launch after it into the L2 of the XCD that will read them
	// ---------------------------------------------------------------------------------------------------------------
	// What the NEXT launch will stream: its workgroup c reads bytes [c * chunkBytes, (c + 1) * chunkBytes) of the region, and
	// workgroup c runs on XCD c % 8 (dispatch order; a wrong guess costs the L2 hit, never correctness).
	struct PrefetchHint
	{
		const void* ptr;
		long long bytes;
		int chunkBytes;
	};
	constexpr int SMALL_MAX_ROWS = 4;		// activation rows (sequences) of the small-batch decode path
	constexpr int CROSS_SPLITS = 8;			// workgroups that share the keys of one (sequence, head) in cross-attention
	constexpr int CROSS_PART = 68;			// floats per partial result: 64 dims of sum( e * V ), the double sum( e ), padding
	struct SmallGemvArgs
	{
		GemmArgs g;			  // W, M <= 4, N, K, epi and its operands; pro 0: A (FP16 rows, lda); pro 1: lnX / lnW / lnB (row length K)
		int pro;			  // 0 = FP16 activations, 1 = LayerNorm prologue, 3 = cross-attention partials combined into the activations
		const float* part;	  // pro 3: [M][H][CROSS_SPLITS][CROSS_PART]
		PrefetchHint pf[ 2 ];
	};
	int launchGemvSmall( const SmallGemvArgs& a, hipStream_t stream );
	// Cross-attention of single-token steps in two launches of H x CROSS_SPLITS x sequences workgroups:
	//   scores   LayerNorm of the residual row, this head's query (as attentionDecG's fused query), K . q for the split's keys
	//            -> scores[seq][h][key], splitMax[seq][h][split]
	//   softmaxPV  e = exp16( score - max over all splits ), sum( e ) in double, sum( e * V ) -> part[seq][h][split]
	// The consumer (launchGemvSmall, pro 3) adds the splits in order and scales by float( 1 / sum ): the table softmax of the
	// reference with its GLOBAL maximum (ggml.c:5030-5090), only the FP32 summation order differs from attentionDecG.
	struct CrossSplitArgs
	{
		const float* lnX;	  // [seq][d] residual stream
		const float* lnW;
		const float* lnB;
		const f16* qW;		  // [d][d]
		const float* qB;
		float qScale;
		const f16* kc;		  // [seq][H][keyStride][64]
		const f16* vc;
		float* scores;		  // [seq][H][keyStride]
		float* splitMax;	  // [seq][H][CROSS_SPLITS]
		float* part;		  // [seq][H][CROSS_SPLITS][CROSS_PART]
		int batch, H, nKeys, keyStride;
		PrefetchHint pf[ 2 ];
	};
	int launchCrossScores( const CrossSplitArgs& a, hipStream_t stream );
	int launchCrossSoftmaxPV( const CrossSplitArgs& a, hipStream_t stream );

	// ---------------------------------------------------------------------------------------------------------------
	// logits -> probabilities -> greedy token, on the device
	// ---------------------------------------------------------------------------------------------------------------
	struct TokenData
	{
		int id, tid;
		float p, pt, ptsum;
	};
	// probs[row] = table softmax( logits[row] ) (ggml.c:5030-5090)
	int launchVocabSoftMax( const float* logits, float* probs, int rows, int nVocab, hipStream_t stream );
	// ContextImpl::sampleBest on the device
	int launchSampleBest( const float* probs, int rows, int nVocab, int tokenBeg, int tokenSot, int tokenSolm, int tokenNot,
		int forceTimestamp, int isInitial, TokenData* out, hipStream_t stream );

	// beam search on hypothesis groups: the `width` (<= 8) best continuations of every sequence under sampleBest's rules (candidate 0 = sampleBest's
	// pick) -> out [rows][width]; and the self-attention cache rows [0, rows) of sequence parents[j] copied to sequence j (all layers, through a scratch
	// copy of the caches)
	int launchBeamCandidates( const float* probs, int rows, int nVocab, int tokenBeg, int tokenSot, int tokenSolm, int tokenNot,
		int forceTimestamp, int isInitial, int width, TokenData* out, hipStream_t stream );
	int launchReorderCache( f16* cacheK, f16* cacheV, f16* scratchK, f16* scratchV, const int* parents, int layers, int sequences, int maxSeq,
		int heads, int keyStride, int rows, hipStream_t stream );

	// ---- beam search WITHOUT the host in the loop (round 5) ------------------------------------------------------------------
	// One window = `slots` decoder sequences sharing its cross-attention K/V. After every decode step beamRankKernel does, per window, what
	// ContextImpl::decodeWindowBeam (host/whisperImpl.cpp) does on the host: pool = live hypotheses x their `width` candidates ranked by
	// parent score + log p (stable: ties keep parent order, then candidate order), the best `width` accepted, each fed through the
	// window's stop rules (WindowScan of host/hostLoop.h = ContextImpl.cpp:597-673, restated below as beamFeed) -> live or finished;
	// the search ends when no live hypothesis can still beat the best finished one. It writes where the NEXT step reads: parents[] (what
	// reorderCacheKernel moves), nextTokens[] (what the embedding reads), and one BeamRecord per accepted proposal (token + the record of its
	// parent) from which the host re-derives the winner's token chain. Everything of a step is in one captured graph; the host polls `done`.
	struct BeamRules
	{
		int seek, seekEnd;		 // WindowScan: the window's first frame and the stream's end, 10 ms units
		int nMax;				 // n_text_ctx / 2 - 4
		int maxTokens;			 // sFullParams::max_tokens (0 = no limit)
		int singleSegment;		 // eFullParamsFlags::SingleSegment
		int tokenBeg, tokenEot;
		int forced;				 // != 0: no stop rules at all -- every hypothesis lives for as many steps as are enqueued (random-weight workloads)
	};
	struct BeamHyp
	{
		double sum;				 // cumulative log-probability
		int i, hasTs, seekDelta, resultLen, failed, over;	 // WindowScan's state
		int nTok;				 // tokens WindowScan kept
		int rec;				 // record of its last token: step * width + index among that step's accepted proposals
	};
	constexpr int BEAM_MAX_WIDTH = 8, BEAM_MAX_FINISHED = 3 * BEAM_MAX_WIDTH;
	struct BeamWindow
	{
		int step;				 // ranking steps done (the first one ranks the prompt step's candidates)
		int nLive, nFinished, done;
		int nPrompt, nTextCtx, reserved0, reserved1;
		BeamHyp live[ BEAM_MAX_WIDTH ];	 // live[ j ] decodes in slot j of the window
		BeamHyp finished[ BEAM_MAX_FINISHED ];
	};
	struct BeamRecord
	{
		TokenData t;
		int parent;				 // record index of the hypothesis it continues (-1: the prompt)
		int finished;			 // 1: this proposal ended its hypothesis' window
		int reserved;
	};
	// cand [windows * slots][width] from beamCandidatesKernel; records [maxSteps][windows][width]; parents / nextTokens [windows * slots] (absolute sequence indices)
	int launchBeamRank( const TokenData* cand, int windows, int slots, int width, const BeamRules* rules, BeamWindow* state, BeamRecord* records, int maxSteps,
		int* parents, int* nextTokens, hipStream_t stream );
	// launchReorderCache with the number of rows taken from device memory: rowsDev[ j ] rows of sequence j (its decoder position)
	// group > 1 (beam search on the device: parents stay inside their window's group of `group` consecutive sequences): one launch through registers, no scratch copy
	int launchReorderCacheDev( f16* cacheK, f16* cacheV, f16* scratchK, f16* scratchV, const int* parents, const int* rowsDev, int layers, int sequences, int maxSeq,
		int heads, int keyStride, int group, hipStream_t stream );

	// Device-resident state of the greedy loop: lets one captured hipGraph be replayed for every token. The POSITIONS live next to
	// it as one int per sequence (wh_context::seqPos): the sequences of a lock-step batch may stand at different positions (streams
	// of a batch scheduler carry prompts of different lengths), every kernel that needs a position reads its own sequence's.
	struct DecodeState
	{
		int step;			 // index of the next sample in the output array
		int forceTimestamp;	 // consumed (cleared) by the sampler
		int isInitial;
		int gen;			 // window generation: stamped into the host mailbox with every sample (0 = no mailbox writes)
	};
	// Pinned, host-coherent mirror of the greedy loop's samples: the sampler writes sample `step` of row r to data[step * rows + r]
	// and then (system-scope fence in between) stamps flag[step * rows + r] = gen. A host thread that polls the flag gets the
	// token without an event, a copy or a stream synchronisation on the decode stream.
	// flag holds two ints per record: [2 * slot] = the stamp, [2 * slot + 1] = a checksum of the record and the generation.
	struct SampleMailbox
	{
		TokenData* data;	 // device-visible address of the pinned array
		int* flag;
	};
	// logits row -> table softmax -> ContextImpl::sampleBest, one kernel; writes TokenData to out[state->step * rows + row],
	// the chosen id to nextTokens[row]; probsOut optional.
	int launchSoftMaxSample( const float* logits, float* probsOut, int rows, int nVocab, int tokenBeg, int tokenSot, int tokenSolm,
		int tokenNot, const DecodeState* state, TokenData* out, int* nextTokens, SampleMailbox mail, hipStream_t stream );
	// the same sampler with every row cut into 64 slices over the chip (three small launches; the few rows of one stream). scratch: sampleSpreadScratchBytes( rows )
	// of device memory; eOut receives the UNNORMALISED exponentials
	size_t sampleSpreadScratchBytes( int rows );
	int launchSoftMaxSampleSpread( const float* logits, float* eOut, int rows, int nVocab, int tokenBeg, int tokenSot, int tokenSolm, int tokenNot,
		const DecodeState* state, TokenData* out, int* nextTokens, SampleMailbox mail, void* scratch, hipStream_t stream );
	// step += 1, the sampler flags cleared, seqPos[0 .. rows) += 1
	int launchAdvanceState( DecodeState* state, int* seqPos, int rows, hipStream_t stream );
	// Prompt steps whose sequences differ in length (rows right-padded to nTok tokens): the normalised row of sequence b's LAST real
	// token, lastPos[b] (device), is copied over its row nTok - 1, where the vocabulary product of a prompt step reads. In place, f16 rows.
	int launchGatherLastRows( f16* xn, const int* lastPos, int batch, int nTok, int d, hipStream_t stream );

	// ---------------------------------------------------------------------------------------------------------------
	// mel spectrogram
	// ---------------------------------------------------------------------------------------------------------------
	// raw log-mel (before clamp/normalise) [nMel][nLen] + running maximum; then normalise in place
	int launchMel( const float* pcm, long long nSamples, const float* filters, const double* dftTable, float* mel, long long nLen,
		int nMel, float* maxScratch, hipStream_t stream );
	// one window of a streamed spectrogram, normalised by its own maximum (MelStreamer semantics); maxScratch holds 2 ints
	int launchMelBatch( const float* pcm, long long nSamples, long long pcmStride, int batch, const float* filters, const double* dftTable, float* mel, long long melStride,
		long long nLen, int nMel, float* maxScratch, hipStream_t stream );	 // 1 = shape not covered (loop over launchMel)
	int launchMelWindow( const float* pcm, long long nSamples, const float* filters, const double* dftTable, float* mel, long long nLen,
		long long nValidFrames, int nMel, int reusePreviousMax, float* maxScratch, hipStream_t stream );
}
