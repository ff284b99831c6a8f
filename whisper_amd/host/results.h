// iTranscribeResult of the library: segments + tokens in the reference's POD layout (Whisper/Whisper/TranscribeResult.h), filled from
// the transcript a stream built (hostLoop.h). Shared by iContext::getResults and Whisper::runFullBatch.
#pragma once
#include "hostCommon.h"
#include "hostLoop.h"

namespace Whisper
{
	struct ResultData
	{
		std::vector<sSegment> segments;
		std::vector<sToken> tokens;
		std::vector<std::string> texts;
	};
	class TranscribeResult : public ComObject<iTranscribeResult>, public ResultData
	{
	public:
		HRESULT getSize( sTranscribeLength& rdi ) const override
		{
			rdi.countSegments = (uint32_t)segments.size();
			rdi.countTokens = (uint32_t)tokens.size();
			return S_OK;
		}
		const sSegment* getSegments() const override { return segments.empty() ? nullptr : segments.data(); }
		const sToken* getTokens() const override { return tokens.empty() ? nullptr : tokens.data(); }
	};

	// The object EMBEDDED in a context: handed out by getResults without eResultFlags::NewObject, it lives as long as the context and its
	// Release never deletes (Whisper/Whisper/TranscribeResult.h:34-43, ContextImpl.misc.cpp:203-209): the idiomatic
	// `CComPtr<iTranscribeResult> r; ctx->getResults( flags, &r );` of a callback (Examples/main/main.cpp:57-58) releases it on scope exit.
	class TranscribeResultStatic : public iTranscribeResult, public ResultData
	{
	public:
		HRESULT QueryInterface( const ComLight::GUID& riid, void** ppv ) override
		{
			if( !ppv ) return E_POINTER;
			if( riid == iTranscribeResult::iid() || riid == ComLight::IID_IUnknown ) { *ppv = this; return S_OK; }
			return E_NOINTERFACE;
		}
		uint32_t AddRef() override { return 1; }
		uint32_t Release() override { return 1; }
		HRESULT getSize( sTranscribeLength& rdi ) const override
		{
			rdi.countSegments = (uint32_t)segments.size();
			rdi.countTokens = (uint32_t)tokens.size();
			return S_OK;
		}
		const sSegment* getSegments() const override { return segments.empty() ? nullptr : segments.data(); }
		const sToken* getTokens() const override { return tokens.empty() ? nullptr : tokens.data(); }
	};

	// Segment times: 10 ms units -> 100 ns ticks, plus the media time of the first sample (ContextImpl.misc.cpp getResults)
	inline HRESULT fillResultData( const std::vector<Segment>& resultAll, const Vocabulary& vocab, int64_t mediaTimeOffset, eResultFlags flags, ResultData& res )
	{
		const bool withTokens = flags & eResultFlags::Tokens, withTimes = flags & eResultFlags::Timestamps;
		res.segments.resize( resultAll.size() );
		res.texts.resize( resultAll.size() );
		size_t tc = 0;
		if( withTokens )
			for( const Segment& s : resultAll ) tc += s.tokens.size();
		res.tokens.resize( tc );
		size_t soFar = 0;
		auto ticks = []( int64_t t10ms ) { return (uint64_t)( t10ms * 100000 ); };	 // 10 ms -> 100 ns
		// A token whose times were never computed (no TokenTimestamps flag) reports 0. Under the GPU model's rules a token's times START
		// at 0 instead of "unknown" (hostLoop.h), so makeResults reports the media time for it (ContextImpl.misc.cpp:277-283); pinned on
		// the reference's own makeResults, tests/golden/ref_hostloop_contextimpl.json "results".
		const bool fromZero = g_hostLoopRules == eHostLoopRules::ContextImpl;
		auto tokenTicks = [ & ]( int64_t t10ms ) -> uint64_t
		{
			if( !withTimes || ( t10ms < 0 && !fromZero ) ) return 0;
			return ticks( std::max<int64_t>( t10ms, 0 ) ) + (uint64_t)mediaTimeOffset;
		};
		for( size_t i = 0; i < resultAll.size(); i++ )
		{
			const Segment& src = resultAll[ i ];
			sSegment& dst = res.segments[ i ];
			res.texts[ i ] = src.text;
			dst.text = res.texts[ i ].c_str();
			dst.time.begin.ticks = withTimes ? ticks( src.t0 ) + (uint64_t)mediaTimeOffset : 0;
			dst.time.end.ticks = withTimes ? ticks( src.t1 ) + (uint64_t)mediaTimeOffset : 0;
			dst.firstToken = (uint32_t)soFar;
			dst.countTokens = (uint32_t)src.tokens.size();
			if( withTokens )
				for( size_t j = 0; j < src.tokens.size(); j++ )
				{
					const TokenData& t = src.tokens[ j ];
					sToken& o = res.tokens[ soFar + j ];
					o.text = vocab.string( t.id );
					o.time.begin.ticks = tokenTicks( t.t0 );
					o.time.end.ticks = tokenTicks( t.t1 );
					o.probability = t.p; o.probabilityTimestamp = t.pt; o.ptsum = t.ptsum; o.vlen = t.vlen;
					o.id = t.id;
					o.flags = t.id >= vocab.token_eot ? eTokenFlags::Special : eTokenFlags::None;
				}
			soFar += src.tokens.size();
		}
		return S_OK;
	}
}
