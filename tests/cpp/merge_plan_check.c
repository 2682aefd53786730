/* Plain C caller of the two host-arithmetic entry points of ABI 4 (no context, no device): pjb_plan_groups and pjb_merge_rows, as the
 * reference's JunctionBuilder would use them around its own all-gather ("Multi-GPU").  Built and run by
 * tests/test_abi_exports.py::test_c_caller_of_plan_groups_and_merge_rows. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "portcullis_amd.h"

#define CHECK(c)                                                      \
    do {                                                              \
        if (!(c)) {                                                   \
            fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c);    \
            return 1;                                                 \
        }                                                             \
    } while (0)

int main(void) {
    /* the chain plan of GRCh38's 25 sequences: three groups of about a gigabase */
    static const int32_t len[25] = {248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
                                    133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,  83257441,  80373285,
                                    58617616,  64444167,  46709983,  50818468,  156040895, 57227415,  16569};
    int32_t tids[25], group_of[25];
    for (int i = 0; i < 25; i++) tids[i] = i;
    CHECK(pjb_plan_groups(len, tids, 25, 0, group_of) == 3);
    CHECK(group_of[0] == 0 && group_of[4] == 0 && group_of[5] == 1 && group_of[11] == 1 && group_of[12] == 2 && group_of[24] == 2);
    CHECK(pjb_plan_groups(len, tids, 0, 0, group_of) == 0);
    CHECK(pjb_plan_groups(NULL, tids, 3, 0, group_of) < 0);

    /* two ranks' send slots as pjb_set_row_mirror leaves them: header (n_rows, spliced, unspliced, sum_len, min_len, max_len, 0, 0) + rows */
    enum { CAP = 4 };
    const size_t stride = PJB_MIRROR_HEADER_BYTES + CAP * sizeof(pjb_junction_row);
    unsigned char *g = (unsigned char *)calloc(2, stride);
    CHECK(g != NULL);
    int64_t h0[8] = {3, 10, 5, 1500, 149, 151, 0, 0}, h1[8] = {2, 7, 1, 800, 150, 152, 0, 0};
    memcpy(g, h0, sizeof h0);
    memcpy(g + stride, h1, sizeof h1);
    pjb_junction_row *r0 = (pjb_junction_row *)(g + PJB_MIRROR_HEADER_BYTES), *r1 = (pjb_junction_row *)(g + stride + PJB_MIRROR_HEADER_BYTES);
    r0[0].refid = 4, r0[0].start = 10;   /* rank 0 finished target 4, then target 1 */
    r0[1].refid = 4, r0[1].start = 20;
    r0[2].refid = 1, r0[2].start = 7;
    r1[0].refid = 2, r1[0].start = 99;   /* rank 1: targets 2 and 0 */
    r1[1].refid = 0, r1[1].start = 5;
    pjb_junction_row out[8];
    int64_t n = -1;
    pjb_region_result tot;
    CHECK(pjb_merge_rows(g, 2, (int64_t)stride, out, 8, &n, &tot) == PJB_OK);
    CHECK(n == 5);
    CHECK(out[0].refid == 0 && out[1].refid == 1 && out[2].refid == 2 && out[3].refid == 4 && out[3].start == 10 && out[4].refid == 4 && out[4].start == 20);
    CHECK(tot.spliced == 17 && tot.unspliced == 6 && tot.sum_len == 2300 && tot.min_len == 149 && tot.max_len == 152 && tot.n_reads == 23 && tot.n_junctions == 5);
    CHECK(pjb_merge_rows(g, 2, (int64_t)stride, out, 4, &n, &tot) == PJB_ERR_ARG && n == 5); /* too little room: says how many */
    free(g);
    printf("ok\n");
    return 0;
}
