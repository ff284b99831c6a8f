// oracle/shim/melstreamer/overlay/vocabulary_table.h -- TEST INFRASTRUCTURE ONLY.
// Linked into the build tree of oracle/Makefile (libcontextimpl_ref.so) under the name Whisper/Whisper/Vocabulary.h. The reference's
// class keeps its strings in an ATL hash map and loads them from a ComLight stream (Vocabulary.cpp); the host loop only READS the
// vocabulary -- the special token ids, the token count and a token's text (ContextImpl.cpp:74-153, :455-507, :700-703;
// ContextImpl.misc.cpp:273, :333-343) -- so here it is a table the harness fills from the same ggml file through the reference's CPU
// model (whisper_token_to_str, whisper_token_*).
#pragma once
#include <string>
#include <vector>
namespace Whisper
{
	class Vocabulary
	{
	public:
		using id = int;
		std::vector<std::string> table;	   // filled by oracle/contextimpl_harness.cpp
		int n_vocab = 51864;
		id token_eot = 50256, token_sot = 50257, token_prev = 50360, token_solm = 50361, token_not = 50362, token_beg = 50363;
		static constexpr id token_translate = 50358, token_transcribe = 50359;	   // (Vocabulary.h:29-30)

		bool is_multilingual() const { return n_vocab == 51865; }
		size_t size() const { return table.size(); }
		const char* string( int i ) const { return i >= 0 && i < (int)table.size() ? table[ i ].c_str() : nullptr; }
	};
}
