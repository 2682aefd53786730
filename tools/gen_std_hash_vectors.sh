#!/bin/bash
# Golden vectors for std::hash<std::string> (GNU libstdc++) -> tests/golden/std_hash_vectors.json.
# The reference keys its spliced-alignment map by this hash of BamAlignment::deriveName()
# (lib/include/portcullis/junction.hpp:158).  Generated with the image's g++ (libstdc++).
set -e
cd "$(dirname "$0")/.."
cat > /tmp/std_hash_vec.cc <<'CC'
#include <cstdio>
#include <functional>
#include <string>
int main() {
    const char* names[] = {"", "a", "ab", "abcdefg", "abcdefgh", "abcdefghi", "read_12345_R1", "read_12345_R2", "read_12345_R?",
                           "HWI-ST1234:88:C1234ACXX:3:1101:1234:2066_R2", "s0000000012", "SRR1234567.98765432_R1",
                           "D00360:94:H2YT5BCXX:1:1101:1219:2228_R1",
                           "a_name_of_exactly_sixty_four_bytes_0123456789012345678901234567"};
    printf("{\n");
    const int n = sizeof(names) / sizeof(names[0]);
    for (int i = 0; i < n; i++) printf(" \"%s\": %zu%s\n", names[i], std::hash<std::string>()(names[i]), i + 1 < n ? "," : "");
    printf("}\n");
}
CC
g++ -O1 -o /tmp/std_hash_vec /tmp/std_hash_vec.cc
/tmp/std_hash_vec > tests/golden/std_hash_vectors.json
cat tests/golden/std_hash_vectors.json
