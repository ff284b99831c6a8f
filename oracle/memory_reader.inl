// oracle/memory_reader.inl -- TEST INFRASTRUCTURE ONLY; included by melstreamer_harness.cpp and contextimpl_harness.cpp (after MelStreamer.h).
// What the reference's spectrogram sources link against and Windows / Media Foundation would provide: PcmReader's three methods over PCM in
// memory (shim/melstreamer/mfidl.h), restating MF/PcmReader.cpp:307-428 statement for statement for a mono source (see "THE END OF A STREAM" in
// melstreamer_harness.cpp for what that reproduces), the thread pool of Utils/parallelFor.h as plain std::threads, setCurrentThreadName.
// ---- PcmReader over memory ----
PcmReader::PcmReader( const iAudioReader* iar )
{
	if( nullptr == iar ) throw E_POINTER;
	check( iar->getReader( &reader ) );
	sampleHandler = nullptr;	// mono source, mono output (PcmReader.cpp:286-288: HandlerMono)
	m_length = reader->count / FFT_STEP;
}
HRESULT PcmReader::readNextSample()
{
	// PcmReader.cpp:307-322 with HandlerMono::moveBufferData (:56-67): the unconsumed tail moves to the front
	const size_t off = bufferReadOffset;
	const size_t available = pcm.mono.size() - off;
	if( available > 0 )
	{
		if( 0 != off )
		{
			memmove( pcm.mono.data(), pcm.mono.data() + off, available * 4 );
			pcm.mono.resize( available );
		}
	}
	else
		pcm.clear();
	bufferReadOffset = 0;
	IMFSourceReader& r = *reader;
	if( r.cursor >= r.count ) return E_EOF;
	const size_t n = std::min( r.block, r.count - r.cursor );
	pcm.appendMono( r.pcm + r.cursor, n );
	r.cursor += n;
	return S_OK;
}
HRESULT PcmReader::readChunk( PcmMonoChunk& mono, PcmStereoChunk* )
{
	while( true )
	{
		const size_t off = bufferReadOffset;
		const size_t available = pcm.mono.size() - off;
		if( available >= FFT_STEP )
		{
			memcpy( mono.mono.data(), &pcm.mono[ off ], FFT_STEP * 4 );
			bufferReadOffset = off + FFT_STEP;
			return S_OK;
		}
		if( !m_readerEndOfFile )
		{
			const HRESULT hr = readNextSample();
			if( SUCCEEDED( hr ) ) continue;
			if( hr != E_EOF ) return hr;
			m_readerEndOfFile = true;
		}
		if( available > 0 )
		{
			memcpy( mono.mono.data(), &pcm.mono[ off ], available * 4 );
			memset( mono.mono.data() + available, 0, ( FFT_STEP - available ) * 4 );
			bufferReadOffset = off + available;
			return S_OK;
		}
		return E_EOF;
	}
}

// ---- ThreadPoolWork: threadPoolCallback( 0 .. n-1 ) on n threads, the first failure is the result ----
ThreadPoolWork::~ThreadPoolWork() {}
HRESULT ThreadPoolWork::create() { return S_OK; }
HRESULT ThreadPoolWork::parallelFor( int threadsCount ) noexcept
{
	std::vector<std::thread> ts;
	std::vector<HRESULT> hrs( (size_t)threadsCount, S_OK );
	for( int i = 1; i < threadsCount; i++ ) ts.emplace_back( [ this, i, &hrs ]() { hrs[ i ] = threadPoolCallback( i ); } );
	hrs[ 0 ] = threadPoolCallback( 0 );
	for( auto& t : ts ) t.join();
	for( HRESULT hr : hrs )
		if( FAILED( hr ) ) return hr;
	return S_OK;
}

// the free function of Utils/parallelFor.h (Spectrogram::pcmToMel with threads >= 2, Spectrogram.cpp:86-93)
HRESULT Whisper::parallelFor( pfnParallelForCallback pfn, int threadsCount, void* ctx )
{
	std::vector<std::thread> ts;
	std::vector<HRESULT> hrs( (size_t)threadsCount, S_OK );
	for( int i = 1; i < threadsCount; i++ ) ts.emplace_back( [ pfn, ctx, i, &hrs ]() { hrs[ i ] = pfn( i, ctx ); } );
	hrs[ 0 ] = pfn( 0, ctx );
	for( auto& t : ts ) t.join();
	for( HRESULT hr : hrs )
		if( FAILED( hr ) ) return hr;
	return S_OK;
}

void setCurrentThreadName( const char* ) {}
