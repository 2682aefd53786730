// The code the reference keys read names by: std::hash<std::string>()(BamAlignment::deriveName())
// (lib/include/portcullis/junction.hpp:158, lib/src/bam_alignment.cc:233-242).  deriveName() is QNAME plus "_R1" /
// "_R2" / "_R?" for paired reads.  std::hash<std::string> of GNU libstdc++ is _Hash_bytes(ptr, len, 0xc70f6907)
// (libstdc++-v3/libsupc++/hash_bytes.cc, 64-bit: a MurmurHash64A variant); restated here so that the record
// transcoder does not build strings.  Only equality of codes reaches the output (mm_score).
#pragma once
#include <cstdint>

namespace portcullis {
namespace bam {

inline uint64_t deriveNameHash(const uint8_t* name, uint32_t len, uint32_t flag) {
    const uint64_t mul = (((uint64_t)0xc6a4a793UL) << 32) + (uint64_t)0x5bd1e995UL;
    uint8_t suf[3] = {'_', 'R', '?'};
    uint32_t total = len;
    if (flag & 0x1u) {
        suf[2] = (flag & 0x40u) ? '1' : (flag & 0x80u) ? '2' : '?';
        total += 3;
    }
    auto at = [&](uint32_t i) -> uint64_t { return i < len ? name[i] : suf[i - len]; };
    auto mix = [](uint64_t v) { return v ^ (v >> 47); };
    uint64_t hash = 0xc70f6907ULL ^ ((uint64_t)total * mul);
    const uint32_t aligned = total & ~7u;
    for (uint32_t p = 0; p < aligned; p += 8) {
        uint64_t w = 0;
        for (int k = 7; k >= 0; k--) w = (w << 8) | at(p + (uint32_t)k);
        hash ^= mix(w * mul) * mul;
        hash *= mul;
    }
    if (total & 7u) {
        uint64_t data = 0;
        for (int n = (int)(total & 7u) - 1; n >= 0; n--) data = (data << 8) + at(aligned + (uint32_t)n);
        hash ^= data;
        hash *= mul;
    }
    hash = mix(hash) * mul;
    return mix(hash);
}

}  // namespace bam
}  // namespace portcullis
