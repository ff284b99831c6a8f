// Token-level timestamps and max_len wrapping: host-only post-processing of a finished segment.
//
// Restates whisper_exp_compute_token_level_timestamps / whisper_wrap_segment of the reference's CPU model
// (Whisper/source/whisper.cpp:3320-3575, 2711-2760; Const-me's copy: Whisper/Whisper/ContextImpl.cpp:219-419) with the same
// integer and floating-point arithmetic, so that the token times are the reference's for the same token data; the pins are
// outputs of oracle/_ref (tests/golden/ref_token_timestamps.json).
#include "hostCommon.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace Whisper
{
	namespace
	{
		constexpr int SAMPLE_RATE = 16000;

		// whisper.cpp:3322-3328; times are 10 ms units
		int sampleOfTime( int64_t t, int nSamples )
		{
			return std::max( 0, std::min( nSamples - 1, (int)( ( t * SAMPLE_RATE ) / 100 ) ) );
		}
		int64_t timeOfSample( int sample ) { return ( (int64_t)100 * sample ) / SAMPLE_RATE; }

		// heuristic cost of pronouncing a token (whisper.cpp:3332-3354)
		float voiceLength( const char* text )
		{
			float res = 0.0f;
			for( const char* p = text; p && *p; p++ )
			{
				const char c = *p;
				if( c == ' ' ) res += 0.01f;
				else if( c == ',' ) res += 2.00f;
				else if( c == '.' || c == '!' || c == '?' ) res += 3.00f;
				else if( c >= '0' && c <= '9' ) res += 3.00f;
				else res += 1.00f;
			}
			return res;
		}
	}

	void TokenTimestamper::begin( const float* pcm, size_t samples )
	{
		tBeg = tLast = 0;
		tidLast = 0;
		// mean of |x| over [i - 32, i + 32] clipped to the signal, divided by 65 regardless (whisper.cpp:3357-3373);
		// summed in the reference's order so that threshold comparisons further down see the same floats
		constexpr int hw = 32;
		const int n = (int)samples;
		energy.assign( samples, 0.0f );
		for( int i = 0; i < n; i++ )
		{
			float sum = 0;
			const int j0 = std::max( i - hw, 0 ), j1 = std::min( i + hw, n - 1 );
			for( int j = j0; j <= j1; j++ ) sum += std::fabs( pcm[ j ] );
			energy[ i ] = sum / ( 2 * hw + 1 );
		}
	}

	void TokenTimestamper::compute( Segment& segment, const Vocabulary& vocab, float tholdPt, float tholdPtsum )
	{
		std::vector<TokenData>& tokens = segment.tokens;
		const int nSamples = (int)energy.size();
		if( nSamples == 0 )
		{
			logWarning( "token timestamps: no signal data available" );
			return;
		}
		const int64_t t0 = segment.t0, t1 = segment.t1;
		const int n = (int)tokens.size();
		if( n == 0 ) return;
		if( n == 1 )
		{
			tokens[ 0 ].t0 = t0;
			tokens[ 0 ].t1 = t1;
			return;
		}

		// 1. anchors: a timestamp candidate (tid) the model is confident about and that moves forward fixes the boundary
		//    between the previous token and this one (whisper.cpp:3408-3443)
		for( int j = 0; j < n; j++ )
		{
			TokenData& tok = tokens[ j ];
			if( j == 0 )
			{
				if( tok.id == vocab.token_beg )
				{
					tokens[ 0 ].t0 = t0;
					tokens[ 0 ].t1 = t0;
					tokens[ 1 ].t0 = t0;
					tBeg = t0;
					tLast = t0;
					tidLast = vocab.token_beg;
				}
				else
					tokens[ 0 ].t0 = tLast;
			}
			const int64_t tt = tBeg + 2 * ( tok.tid - vocab.token_beg );
			tok.vlen = voiceLength( vocab.string( tok.id ) );
			if( tok.pt > tholdPt && tok.ptsum > tholdPtsum && tok.tid > tidLast && tt <= t1 )
			{
				if( j > 0 ) tokens[ j - 1 ].t1 = tt;
				tok.t0 = tt;
				tidLast = tok.tid;
			}
		}
		tokens[ n - 2 ].t1 = t1;
		tokens[ n - 1 ].t0 = t1;
		tokens[ n - 1 ].t1 = t1;
		tLast = t1;

		// 2. runs of tokens without an end time share their interval in proportion to their voice lengths (:3450-3490)
		for( int p0 = 0, p1 = 0;; )
		{
			while( p1 < n && tokens[ p1 ].t1 < 0 ) p1++;
			if( p1 >= n ) p1--;
			if( p1 > p0 )
			{
				double psum = 0.0;
				for( int j = p0; j <= p1; j++ ) psum += tokens[ j ].vlen;
				const double dt = (double)( tokens[ p1 ].t1 - tokens[ p0 ].t0 );
				for( int j = p0 + 1; j <= p1; j++ )
				{
					const double ct = (double)tokens[ j - 1 ].t0 + dt * tokens[ j - 1 ].vlen / psum;
					tokens[ j - 1 ].t1 = (int64_t)ct;
					tokens[ j ].t0 = (int64_t)ct;
				}
			}
			p1++;
			p0 = p1;
			if( p1 >= n ) break;
		}

		// 3. monotonic fix-up (:3493-3505)
		for( int j = 0; j < n - 1; j++ )
		{
			if( tokens[ j ].t1 < 0 ) tokens[ j + 1 ].t0 = tokens[ j ].t1;
			if( j > 0 && tokens[ j - 1 ].t1 > tokens[ j ].t0 )
			{
				tokens[ j ].t0 = tokens[ j - 1 ].t1;
				tokens[ j ].t1 = std::max( tokens[ j ].t0, tokens[ j ].t1 );
			}
		}

		// 4. voice activity: move each text token's edges to where the smoothed energy crosses half of its local mean (:3509-3570)
		const int hw = SAMPLE_RATE / 8;
		for( int j = 0; j < n; j++ )
		{
			if( tokens[ j ].id >= vocab.token_eot ) continue;
			int s0 = sampleOfTime( tokens[ j ].t0, nSamples );
			int s1 = sampleOfTime( tokens[ j ].t1, nSamples );
			const int ss0 = std::max( s0 - hw, 0 );
			const int ss1 = std::min( s1 + hw, nSamples );
			const int ns = ss1 - ss0;
			float sum = 0.0f;
			for( int k = ss0; k < ss1; k++ ) sum += energy[ k ];
			const float thold = (float)( 0.5 * sum / ns );
			{
				int k = s0;
				if( energy[ k ] > thold && j > 0 )
				{
					while( k > 0 && energy[ k ] > thold ) k--;
					tokens[ j ].t0 = timeOfSample( k );
					if( tokens[ j ].t0 < tokens[ j - 1 ].t1 )
						tokens[ j ].t0 = tokens[ j - 1 ].t1;
					else
						s0 = k;
				}
				else
				{
					while( energy[ k ] < thold && k < s1 ) k++;
					s0 = k;
					tokens[ j ].t0 = timeOfSample( k );
				}
			}
			{
				int k = s1;
				if( energy[ k ] > thold )
				{
					while( k < nSamples - 1 && energy[ k ] > thold ) k++;
					tokens[ j ].t1 = timeOfSample( k );
					// (sic) the reference bounds this test by the window's sample count, not by the token count
					if( j < ns - 1 && j + 1 < n && tokens[ j ].t1 > tokens[ j + 1 ].t0 )
						tokens[ j ].t1 = tokens[ j + 1 ].t0;
					else
						s1 = k;
				}
				else
				{
					while( energy[ k ] < thold && k > s0 ) k--;
					s1 = k;
					tokens[ j ].t1 = timeOfSample( k );
				}
			}
		}
	}

	int TokenTimestamper::wrapLast( std::vector<Segment>& all, const Vocabulary& vocab, int maxLen )
	{
		if( all.empty() ) return 0;
		Segment segment = all.back();	 // working copy of the piece being scanned
		int pieces = 1, acc = 0;
		std::string text;
		for( int i = 0; i < (int)segment.tokens.size(); i++ )
		{
			const TokenData& tok = segment.tokens[ i ];
			if( tok.id >= vocab.token_eot ) continue;
			const char* const txt = vocab.string( tok.id );
			const int cur = (int)strlen( txt ? txt : "" );
			if( acc + cur > maxLen && i > 0 )
			{
				// close the current piece before token i, open a new one that starts with it
				Segment& done = all.back();
				done.text = std::move( text );
				done.t1 = tok.t0;
				done.tokens.resize( i );
				Segment next;
				next.t0 = tok.t0;
				next.t1 = segment.t1;
				next.tokens.assign( segment.tokens.begin() + i, segment.tokens.end() );
				all.push_back( std::move( next ) );
				acc = 0;
				text.clear();
				segment = all.back();
				i = -1;
				pieces++;
			}
			else
			{
				acc += cur;
				text += txt ? txt : "";
			}
		}
		all.back().text = std::move( text );
		return pieces;
	}
}
