// A raw-DEFLATE decoder for whole BGZF blocks (RFC 1951), written for the one case the host reader has: the complete
// compressed payload of a block in memory, the exact inflated size known from the block's ISIZE field, many blocks decoded
// by many threads into neighbouring ranges of one buffer.  It replaces zlib's inflate() in BamReader's block loops
// (the reference: bgzf_read_block / inflate_block, deps/htslib-1.3/bgzf.c:292-316, 421-540), where zlib's byte-at-a-time
// state machine was 0.38 of the 0.45 s `bamfilt` spent reading the configs[1] file on 16 threads.
//   * 64-bit bit buffer refilled eight bytes at a time; one refill per symbol pair (a match needs at most 48 bits);
//   * 11-bit literal/length table and 8-bit distance table with second-level tables for the longer codes, entries carrying
//     base value and extra-bit count, so a symbol is one lookup;
//   * matches copied eight bytes a step where that cannot touch a byte outside [out, out + outLen) -- the neighbouring
//     ranges belong to other threads.
// It accepts exactly the streams zlib accepts and produces the same bytes (tests/test_host_fast_inflate.py: zlib's own
// output at every level and strategy, stored blocks, truncated and bit-flipped streams, under ASan + UBSan); whatever it
// declines -- damaged input, a code zlib would call incomplete or over-subscribed -- the callers hand to zlib, whose verdict
// (and message) stands.
#pragma once

#include <cstddef>
#include <cstdint>

namespace portcullis {
namespace bam {

// true: the stream at [in, in + inLen) is a complete raw DEFLATE stream and inflated to exactly outLen bytes at out.
// false: anything else (nothing outside [out, out + outLen) was written, nothing outside [in, in + inLen) was read).
bool fastInflate(const uint8_t* in, size_t inLen, uint8_t* out, size_t outLen);

}  // namespace bam
}  // namespace portcullis
