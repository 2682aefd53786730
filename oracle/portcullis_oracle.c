/*
 * portcullis_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see portcullis_oracle.h).
 *
 * Literal plain-C restatement of the reference `junc` path.  Every function
 * cites the reference file:line it follows (paths relative to /root/reference).
 * The restatement deliberately keeps the reference's structure -- strings are
 * built and compared character by character, junctions are grouped through a
 * hash map in first-seen order -- so that it fails differently from the HIP
 * path if either is wrong.
 */
#include "portcullis_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[1024];

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

const char *orc_last_error(void) { return g_err; }
size_t orc_sizeof_row(void) { return sizeof(orc_row); }

/* ------------------------------------------------------------------ */
/* small string builder                                               */
/* ------------------------------------------------------------------ */
typedef struct {
    char *p;
    size_t n, cap;
} sbuf;

static int sb_reserve(sbuf *s, size_t extra) {
    if (s->n + extra + 1 <= s->cap) return 0;
    size_t nc = s->cap ? s->cap * 2 : 256;
    while (nc < s->n + extra + 1) nc *= 2;
    char *np = (char *)realloc(s->p, nc);
    if (!np) return -1;
    s->p = np;
    s->cap = nc;
    return 0;
}
static int sb_put(sbuf *s, const char *d, size_t n) {
    if (sb_reserve(s, n)) return -1;
    memcpy(s->p + s->n, d, n);
    s->n += n;
    s->p[s->n] = 0;
    return 0;
}
static int sb_fill(sbuf *s, char c, size_t n) {
    if (sb_reserve(s, n)) return -1;
    memset(s->p + s->n, c, n);
    s->n += n;
    s->p[s->n] = 0;
    return 0;
}
static int sb_printf(sbuf *s, const char *fmt, ...) {
    va_list ap, ap2;
    va_start(ap, fmt);
    va_copy(ap2, ap);
    int k = vsnprintf(NULL, 0, fmt, ap);
    va_end(ap);
    if (k < 0 || sb_reserve(s, (size_t)k)) {
        va_end(ap2);
        return -1;
    }
    vsnprintf(s->p + s->n, (size_t)k + 1, fmt, ap2);
    va_end(ap2);
    s->n += (size_t)k;
    return 0;
}
static void sb_free(sbuf *s) {
    free(s->p);
    s->p = NULL;
    s->n = s->cap = 0;
}

/* ------------------------------------------------------------------ */
/* CIGAR helpers (lib/include/portcullis/bam/bam_alignment.hpp:44-99) */
/* ------------------------------------------------------------------ */
static const char CIGAR_CHARS[] = "MIDNSHP=XB??????";
static inline char op_chr(uint32_t c) { return CIGAR_CHARS[c & 0xf]; }
static inline int32_t op_len(uint32_t c) { return (int32_t)(c >> 4); }

/* CigarOp::opConsumesQuery, bam_alignment.hpp:75-86 */
static inline int consumes_query(char op) {
    return op == 'M' || op == 'I' || op == 'S' || op == '=' || op == 'X';
}
/* CigarOp::opConsumesReference, bam_alignment.hpp:88-99 */
static inline int consumes_ref(char op) {
    return op == 'M' || op == 'D' || op == 'N' || op == '=' || op == 'X';
}

static inline char up(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

/* ------------------------------------------------------------------ */
/* SeqUtils (lib/include/portcullis/seq_utils.hpp)                    */
/* ------------------------------------------------------------------ */

/* SeqUtils::hammingDistance, seq_utils.hpp:62-77: throws on size mismatch, upper-cases both. */
int orc_hamming(const char *a, size_t na, const char *b, size_t nb) {
    if (na != nb)
        return fail(ORC_ERR_HAMMING_LEN, "Can't find hamming distance of strings that are not the same length (%zu vs %zu)", na, nb);
    int sum = 0;
    for (size_t i = 0; i < na; i++)
        if (up(a[i]) != up(b[i])) sum++;
    return sum;
}

/* REVCOMP_LOOKUP, seq_utils.hpp:33-40.  Index = c - 'A'.  Characters outside
 * 'A'..'Z' index out of bounds in the reference (undefined behaviour); the
 * restatement returns NUL for them. */
static const char REVCOMP_LOOKUP[26] = {'T', 0, 'G', 'H', 0, 0, 'C', 'D', 0, 0, 0, 0, 'K',
                                        'N', 0, 0,   0,   'Y', 'W', 'A', 'A', 'B', 'S', 'X', 'R', 0};

/* SeqUtils::reverseComplement, seq_utils.hpp:110-118 */
void orc_revcomp(const char *in, size_t n, char *out) {
    for (size_t i = 0; i < n; i++) {
        int k = (int)(unsigned char)in[i] - 65;
        char c = (k >= 0 && k < 26) ? REVCOMP_LOOKUP[k] : 0;
        out[n - 1 - i] = c;
    }
}

/* ------------------------------------------------------------------ */
/* Intron (lib/src/intron.cc)                                         */
/* ------------------------------------------------------------------ */

/* Intron::minAnchorLength, intron.cc:67-83 */
int64_t orc_min_anchor(int32_t start, int32_t end, int32_t left, int32_t right) {
    if (left > start)
        return fail(ORC_ERR_MIN_ANCHOR, "The intron start position must be greater than the left anchor start position: %d **** %d-%d **** %d", left, start, end, right);
    if (right < end)
        return fail(ORC_ERR_MIN_ANCHOR, "The intron end position must be less than the right anchor end position: (%d **** %d-%d **** %d)", left, start, end, right);
    int32_t l = start - left, r = right - end;
    return (int64_t)(uint32_t)(l < r ? l : r);
}

/* ------------------------------------------------------------------ */
/* BamAlignment padded sequences (lib/src/bam_alignment.cc)           */
/* ------------------------------------------------------------------ */

/* BamAlignment::getQuerySeqAfterClipping(seq), bam_alignment.cc:256-264.
 * Returns offset/length of the substr (std::string::substr semantics, size_t
 * arithmetic including the "+ 1"). */
static int clip_query(const uint32_t *cigar, int n_cigar, size_t qsize, size_t *off, size_t *len) {
    int32_t dS = 0, dE = 0;
    if (n_cigar > 0) {
        if (op_chr(cigar[0]) == 'S') dS = op_len(cigar[0]);
        if (op_chr(cigar[n_cigar - 1]) == 'S') dE = op_len(cigar[n_cigar - 1]);
    }
    if ((size_t)dS > qsize) return fail(ORC_ERR_CLIP_RANGE, "basic_string::substr: pos %d > size %zu", dS, qsize);
    size_t count = qsize - (size_t)dS - (size_t)dE + 1; /* may wrap, as in the reference */
    size_t avail = qsize - (size_t)dS;
    *off = (size_t)dS;
    *len = count < avail ? count : avail;
    return 0;
}

/* BamAlignment::getPaddedQuerySeq, bam_alignment.cc:341-403 (include_soft_clips=false) */
static int padded_query(const uint32_t *cigar, int n_cigar, int32_t position, int32_t aligned_len,
                        const char *query_seq, size_t qsize, int32_t start, int32_t end,
                        int32_t *actual_start, int32_t *actual_end, sbuf *out) {
    int32_t getEnd = position + aligned_len - 1;
    if (start > getEnd || end < position)
        return fail(ORC_ERR_NO_PRESENCE, "Found an alignment that does not have a presence in the requested region");
    int32_t qPos = 0, rPos = position;
    size_t coff = 0, clen = 0;
    int rc = clip_query(cigar, n_cigar, qsize, &coff, &clen);
    if (rc) return rc;
    const char *query = query_seq + coff;
    out->n = 0;
    if (sb_reserve(out, 1)) return fail(ORC_ERR_NOMEM, "oom");
    out->p[0] = 0;
    for (int k = 0; k < n_cigar; k++) {
        char type = op_chr(cigar[k]);
        int32_t length = op_len(cigar[k]);
        int cRef = consumes_ref(type);
        int cQry = consumes_query(type) && type != 'S';
        if (rPos < start) { /* :353-357 whole op skipped */
            if (cRef) rPos += length;
            if (cQry) qPos += length;
            continue;
        }
        if ((rPos > end && type != 'I') || (type == 'N' && rPos + length > end)) break; /* :359 */
        if (cQry) {
            int32_t len = (rPos + length > end && type != 'I') ? end - rPos + 1 : length; /* :362 */
            if (len == 0)
                return fail(ORC_ERR_ZERO_LEN_OP, "Can't extract cigar op sequence from query string when length has been calculated as 0.");
            if (qPos < 0 || qPos + len > (int32_t)clen)
                return fail(ORC_ERR_QUERY_RANGE, "Can't extract cigar op sequence from query string. qPos=%d len=%d query.size=%zu", qPos, len, clen);
            /* query.substr(qPos, len): len<0 converts to huge count -> to end of string */
            size_t take = len < 0 ? clen - (size_t)qPos : (size_t)len;
            if (sb_put(out, query + qPos, take)) return fail(ORC_ERR_NOMEM, "oom");
        } else if (cRef) { /* D or N: pad with 'X' (BAM_CIGAR_DIFF_CHAR), :390-396 */
            uint32_t len = (rPos + length > end) ? (uint32_t)(end - rPos + 1) : (uint32_t)length;
            if (sb_fill(out, 'X', len)) return fail(ORC_ERR_NOMEM, "oom");
        }
        if (cRef) rPos += length;
        if (cQry) qPos += length;
    }
    *actual_start = position > start ? position : start; /* :400 */
    *actual_end = rPos <= end ? rPos - 1 : end;           /* :401 */
    return 0;
}

/* BamAlignment::getPaddedGenomeSeq, bam_alignment.cc:405-462 (include_soft_clips=false) */
static int padded_genome(const uint32_t *cigar, int n_cigar, int32_t position, int32_t aligned_len,
                         const char *genome_seq, size_t gsize, int32_t start, int32_t end,
                         int32_t q_start, int32_t q_end, sbuf *out) {
    int32_t getEnd = position + aligned_len - 1;
    if (start > getEnd || end < position)
        return fail(ORC_ERR_NO_PRESENCE, "Found an alignment that does not have a presence in the requested region");
    int32_t rPos = position;
    if (q_start - start < 0)
        return fail(ORC_ERR_QREGION, "Query start position was before genomic region start position.  Query start: %d; Genomic start: %d", q_start, start);
    if (end - q_end < 0)
        return fail(ORC_ERR_QREGION, "Query end position was beyond genomic region end position.  Query end: %d; Genomic end: %d", q_end, end);
    out->n = 0;
    if (sb_reserve(out, 1)) return fail(ORC_ERR_NOMEM, "oom");
    out->p[0] = 0;
    for (int k = 0; k < n_cigar; k++) {
        char type = op_chr(cigar[k]);
        int32_t length = op_len(cigar[k]);
        int cRef = consumes_ref(type);
        int cQry = consumes_query(type) && type != 'S';
        if (rPos < q_start) { /* :427-431 */
            if (cRef) rPos += length;
            continue;
        }
        if (rPos > q_end && type != 'I') break; /* :433 */
        if (cRef) {
            int32_t seqOffset = rPos - start;
            int32_t len = rPos + length > q_end ? q_end - rPos + 1 : length;
            if (seqOffset < 0 || seqOffset + len > (int32_t)gsize)
                return fail(ORC_ERR_GENOME_RANGE, "Can't extract cigar op sequence from extracted genome region. offset=%d len=%d region=%zu", seqOffset, len, gsize);
            size_t take = len < 0 ? gsize - (size_t)seqOffset : (size_t)len;
            if (sb_put(out, genome_seq + seqOffset, take)) return fail(ORC_ERR_NOMEM, "oom");
        } else if (cQry) { /* 'I' */
            if (sb_fill(out, 'X', (size_t)length)) return fail(ORC_ERR_NOMEM, "oom");
        }
        if (cRef) rPos += length;
    }
    return 0;
}

int orc_padded_query_seq(const uint32_t *cigar, int n_cigar, int32_t position, int32_t aligned_len,
                         const char *query, int32_t start, int32_t end, int32_t *actual_start,
                         int32_t *actual_end, char *out, size_t out_cap) {
    sbuf s = {0};
    int rc = padded_query(cigar, n_cigar, position, aligned_len, query, strlen(query), start, end,
                          actual_start, actual_end, &s);
    if (rc == 0) {
        if (s.n + 1 > out_cap) rc = fail(ORC_ERR_ARG, "output buffer too small");
        else {
            memcpy(out, s.p, s.n + 1);
            rc = (int)s.n;
        }
    }
    sb_free(&s);
    return rc;
}

int orc_padded_genome_seq(const uint32_t *cigar, int n_cigar, int32_t position, int32_t aligned_len,
                          const char *genome_seq, int32_t start, int32_t end, int32_t q_start,
                          int32_t q_end, char *out, size_t out_cap) {
    sbuf s = {0};
    int rc = padded_genome(cigar, n_cigar, position, aligned_len, genome_seq, strlen(genome_seq),
                           start, end, q_start, q_end, &s);
    if (rc == 0) {
        if (s.n + 1 > out_cap) rc = fail(ORC_ERR_ARG, "output buffer too small");
        else {
            memcpy(out, s.p, s.n + 1);
            rc = (int)s.n;
        }
    }
    sb_free(&s);
    return rc;
}

/* ------------------------------------------------------------------ */
/* Junction pieces (lib/src/junction.cc)                              */
/* ------------------------------------------------------------------ */

static int str4eq(const char *s, const char *lit) { return memcmp(s, lit, 4) == 0; }

/* Junction::hasCanonicalSpliceSites :289-304, predictedStrandFromSpliceSites :306-326,
 * setDonorAndAcceptorMotif :504-516.  CANONICAL_SEQ etc: junction.hpp:73-79. */
int orc_donor_acceptor(const char *seq1, size_t n1, const char *seq2, size_t n2, int read_strand,
                       int *ss_strand, int *cons_strand, uint8_t da1[2], uint8_t da2[2]) {
    if (n1 != 2 || n2 != 2)
        return fail(ORC_ERR_SPLICE_SITE_LEN, "Can't test for valid donor / acceptor when either string are not of length two");
    char seq[4] = {seq1[0], seq1[1], seq2[0], seq2[1]};
    /* reverse complements: GTAG->CTAC, ATAC->GTAT, GCAG->CTGC */
    int css;
    if (str4eq(seq, "GTAG") || str4eq(seq, "CTAC")) css = ORC_CSS_CANONICAL;
    else if (str4eq(seq, "ATAC") || str4eq(seq, "GTAT") || str4eq(seq, "GCAG") || str4eq(seq, "CTGC")) css = ORC_CSS_SEMI;
    else css = ORC_CSS_NO;
    int ss;
    if (str4eq(seq, "GTAG")) ss = ORC_STRAND_POS;
    else if (str4eq(seq, "CTAC")) ss = ORC_STRAND_NEG;
    else if (str4eq(seq, "ATAC") || str4eq(seq, "GCAG")) ss = ORC_STRAND_POS;
    else if (str4eq(seq, "GTAT") || str4eq(seq, "CTGC")) ss = ORC_STRAND_NEG;
    else ss = ORC_STRAND_UNK;
    int cons = read_strand == ss ? read_strand
               : read_strand == ORC_STRAND_UNK ? ss
               : ss == ORC_STRAND_UNK ? read_strand
                                      : ORC_STRAND_UNK;
    if (cons == ORC_STRAND_NEG) {
        orc_revcomp(seq2, 2, (char *)da1);
        orc_revcomp(seq1, 2, (char *)da2);
    } else {
        memcpy(da1, seq1, 2);
        memcpy(da2, seq2, 2);
    }
    *ss_strand = ss;
    *cons_strand = cons;
    return css;
}

/* Junction::calcEntropy(const vector<int32_t>), junction.cc:730-749 */
double orc_entropy(const int32_t *p, size_t n) {
    if (n <= 1) return 0;
    double sum = 0.0;
    int32_t lastOffset = p[0];
    uint32_t readsAtOffset = 0;
    for (size_t i = 0; i < n; i++) {
        int32_t pos = p[i];
        readsAtOffset++;
        if (pos != lastOffset || i == n - 1) {
            double pI = (double)readsAtOffset / (double)n;
            sum += pI * log2(pI);
            lastOffset = pos;
            readsAtOffset = 0;
        }
    }
    return fabs(sum);
}

/* substr helper with std::string semantics */
static void substr(const char *s, size_t n, size_t pos, size_t cnt, const char **o, size_t *on) {
    if (pos > n) pos = n; /* callers never exceed; keep defined */
    size_t a = n - pos;
    *o = s + pos;
    *on = cnt < a ? cnt : a;
}

/* Junction::calcHammingScores, junction.cc:823-857 */
int orc_hamming_scores(const char *la0, size_t nla, const char *li0, size_t nli, const char *ri0,
                       size_t nri, const char *ra0, size_t nra, int cons_strand, uint32_t *h5,
                       uint32_t *h3) {
    int32_t leftDelta = (int32_t)(nla - nri);
    int32_t leftOffset = leftDelta <= 0 ? 0 : leftDelta;
    uint32_t leftLen = (uint32_t)(nla < nri ? nla : nri);
    uint32_t rightLen = (uint32_t)(nli < nra ? nli : nra);
    const char *la, *li, *ri, *ra;
    size_t zla, zli, zri, zra;
    if (nla > leftLen) substr(la0, nla, (size_t)leftOffset, leftLen, &la, &zla);
    else { la = la0; zla = nla; }
    if (nli > rightLen) substr(li0, nli, 0, rightLen, &li, &zli);
    else { li = li0; zli = nli; }
    if (nri > leftLen) substr(ri0, nri, (size_t)leftOffset, leftLen, &ri, &zri);
    else { ri = ri0; zri = nri; }
    if (nra > rightLen) substr(ra0, nra, 0, rightLen, &ra, &zra);
    else { ra = ra0; zra = nra; }
    char a5[16], i5[16], i3[16], a3[16];
    size_t na5, ni5, ni3, na3;
    if (zla > 15 || zli > 15 || zri > 15 || zra > 15) return fail(ORC_ERR_ARG, "hamming window > 15");
    if (cons_strand == ORC_STRAND_NEG) {
        orc_revcomp(ra, zra, a5); na5 = zra;
        orc_revcomp(ri, zri, i5); ni5 = zri;
        orc_revcomp(li, zli, i3); ni3 = zli;
        orc_revcomp(la, zla, a3); na3 = zla;
    } else {
        memcpy(a5, la, zla); na5 = zla;
        memcpy(i5, li, zli); ni5 = zli;
        memcpy(i3, ri, zri); ni3 = zri;
        memcpy(a3, ra, zra); na3 = zra;
    }
    int d5 = orc_hamming(a5, na5, i3, ni3);
    if (d5 < 0) return d5;
    int d3 = orc_hamming(a3, na3, i5, ni5);
    if (d3 < 0) return d3;
    *h5 = (uint32_t)d5;
    *h3 = (uint32_t)d3;
    return 0;
}

/* ------------------------------------------------------------------ */
/* path: per-contig junction building                                 */
/* ------------------------------------------------------------------ */

typedef struct {
    int64_t read; /* index into orc_reads */
    /* AlignmentInfo stats, junction.hpp:138-171 (ctor zeroes all) */
    uint32_t totalUpMatches, totalDownMatches, totalUpMism, totalDownMism;
    uint32_t upMatches, downMatches, minMatch, maxMatch, nbMismatches, mmes;
} alninfo;

typedef struct {
    orc_row r;
    alninfo *al;
    size_t n_al, cap_al;
} junc;

typedef struct {
    int32_t start, end;
    int64_t j; /* -1 empty */
} hslot;

typedef struct {
    int32_t tid, ref_len;
    const char *genome;
    const orc_reads *rd;
    junc *list;
    size_t n, cap;
    hslot *tab;
    size_t tcap;
} jsys;

static uint64_t hkey(int32_t s, int32_t e) {
    uint64_t x = ((uint64_t)(uint32_t)s << 32) | (uint32_t)e;
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

static int tab_grow(jsys *js) {
    size_t nc = js->tcap ? js->tcap * 2 : 1024;
    hslot *nt = (hslot *)malloc(nc * sizeof(hslot));
    if (!nt) return -1;
    for (size_t i = 0; i < nc; i++) nt[i].j = -1;
    for (size_t i = 0; i < js->tcap; i++) {
        if (js->tab[i].j < 0) continue;
        size_t h = hkey(js->tab[i].start, js->tab[i].end) & (nc - 1);
        while (nt[h].j >= 0) h = (h + 1) & (nc - 1);
        nt[h] = js->tab[i];
    }
    free(js->tab);
    js->tab = nt;
    js->tcap = nc;
    return 0;
}

static int64_t tab_find(jsys *js, int32_t s, int32_t e) {
    if (!js->tcap) return -1;
    size_t h = hkey(s, e) & (js->tcap - 1);
    while (js->tab[h].j >= 0) {
        if (js->tab[h].start == s && js->tab[h].end == e) return js->tab[h].j;
        h = (h + 1) & (js->tcap - 1);
    }
    return -1;
}

static int tab_put(jsys *js, int32_t s, int32_t e, int64_t j) {
    if ((js->n + 1) * 2 > js->tcap && tab_grow(js)) return -1;
    size_t h = hkey(s, e) & (js->tcap - 1);
    while (js->tab[h].j >= 0) h = (h + 1) & (js->tcap - 1);
    js->tab[h].start = s;
    js->tab[h].end = e;
    js->tab[h].j = j;
    return 0;
}

static int nb_junctions_in_read(const orc_reads *rd, int64_t i) { /* bam_alignment.cc:303-311 */
    int n = 0;
    for (uint32_t k = rd->cig_off[i]; k < rd->cig_off[i + 1]; k++)
        if (op_chr(rd->cigar[k]) == 'N') n++;
    return n;
}

/* Junction::addJunctionAlignment, junction.cc:477-502 */
static int add_junction_alignment(jsys *js, junc *j, int64_t i) {
    if (j->n_al == j->cap_al) {
        size_t nc = j->cap_al ? j->cap_al * 2 : 4;
        alninfo *na = (alninfo *)realloc(j->al, nc * sizeof(alninfo));
        if (!na) return fail(ORC_ERR_NOMEM, "oom");
        j->al = na;
        j->cap_al = nc;
    }
    alninfo *a = &j->al[j->n_al++];
    memset(a, 0, sizeof *a);
    a->read = i;
    j->r.nb_raw = (uint32_t)j->n_al;
    uint16_t flag = js->rd->flag[i];
    int first = (flag & 0x40) != 0, rev = (flag & 0x10) != 0;
    if (first) { if (!rev) j->r.r1pos++; else j->r.r1neg++; }
    else       { if (!rev) j->r.r2pos++; else j->r.r2neg++; }
    if (nb_junctions_in_read(js->rd, i) > 1) j->r.nb_ms++;
    return 0;
}

/* JunctionSystem::addJunctions, junction_system.cc:140-210 (recursive, as written) */
static int add_junctions(jsys *js, int64_t i, size_t startOp, int32_t offset, int *found) {
    const orc_reads *rd = js->rd;
    const uint32_t *cig = rd->cigar + rd->cig_off[i];
    size_t nbOps = rd->cig_off[i + 1] - rd->cig_off[i];
    int32_t lStart = offset, lEndExc = lStart, rStart = lStart, rEndExc = lStart;
    for (size_t k = startOp; k < nbOps; k++) {
        char type = op_chr(cig[k]);
        int32_t length = op_len(cig[k]);
        if (type == 'N') {
            *found = 1;
            int32_t refLength = js->ref_len;
            rStart = lEndExc + length;
            rEndExc = rStart;
            size_t j = k + 1;
            while (j < nbOps && rEndExc <= refLength && op_chr(cig[j]) != 'N') {
                uint32_t r = cig[j++];
                if (consumes_ref(op_chr(r))) rEndExc += op_len(r);
            }
            if (rStart - 1 >= refLength) rStart = refLength - 1; /* :169-171 */
            if (rEndExc - 1 >= refLength) rEndExc = refLength;   /* :172-174 */
            int32_t istart = lEndExc, iend = rStart - 1;
            int64_t jx = tab_find(js, istart, iend);
            if (jx < 0) {
                /* Junction ctor, junction.cc:328-387: maxMinAnchor = minAnchorLength(...) */
                int64_t mma = orc_min_anchor(istart, iend, lStart, rEndExc - 1);
                if (mma < 0) return (int)mma;
                if (js->n == js->cap) {
                    size_t nc = js->cap ? js->cap * 2 : 256;
                    junc *nl = (junc *)realloc(js->list, nc * sizeof(junc));
                    if (!nl) return fail(ORC_ERR_NOMEM, "oom");
                    js->list = nl;
                    js->cap = nc;
                }
                junc *J = &js->list[js->n];
                memset(J, 0, sizeof *J);
                J->r.refid = js->tid;
                J->r.start = istart;
                J->r.end = iend;
                J->r.left = lStart;
                J->r.right = rEndExc - 1;
                J->r.read_strand = J->r.ss_strand = J->r.cons_strand = ORC_STRAND_UNK;
                J->r.canonical = ORC_CSS_NO;
                J->r.max_min_anc = (uint32_t)mma;
                J->r.hamming5p = J->r.hamming3p = 10;
                int rc = add_junction_alignment(js, J, i);
                if (rc) return rc;
                if (tab_put(js, istart, iend, (int64_t)js->n)) return fail(ORC_ERR_NOMEM, "oom");
                js->n++;
            } else {
                junc *J = &js->list[jx];
                int rc = add_junction_alignment(js, J, i);
                if (rc) return rc;
                /* Junction::extendAnchors, junction.cc:524-529 */
                int32_t oS = lStart, oE = rEndExc - 1;
                if (oS < J->r.left) J->r.left = oS;
                if (oE > J->r.right) J->r.right = oE;
                int64_t mma = orc_min_anchor(J->r.start, J->r.end, oS, oE);
                if (mma < 0) return (int)mma;
                if ((uint32_t)mma > J->r.max_min_anc) J->r.max_min_anc = (uint32_t)mma;
            }
            if (j < nbOps) { /* :199-202 */
                int dummy = 0;
                int rc = add_junctions(js, i, k + 1, rStart, &dummy);
                if (rc) return rc;
                break;
            }
        } else if (consumes_ref(type)) {
            lEndExc += length;
        }
    }
    return 0;
}

static int32_t aligned_length(const orc_reads *rd, int64_t i) { /* bam_alignment.cc:78-88 */
    int32_t a = 0;
    for (uint32_t k = rd->cig_off[i]; k < rd->cig_off[i + 1]; k++)
        if (consumes_ref(op_chr(rd->cigar[k]))) a += op_len(rd->cigar[k]);
    return a;
}

/* read strand: BamAlignment::init :89-99 -> XS tag else calcStrand(), which
 * returns UNKNOWN because the reader's alignments carry Strandedness::UNKNOWN
 * (bam_alignment.cc:154-165, bam_reader.hpp:68). */
static int read_strand(const orc_reads *rd, int64_t i) {
    switch (rd->xs[i]) {
    case 1: return ORC_STRAND_POS;
    case 2: return ORC_STRAND_NEG;
    default: return ORC_STRAND_UNK;
    }
}

/* BamAlignment::calcIfProperPair, bam_alignment.cc:271-292 */
static int calc_if_proper_pair(const orc_reads *rd, int64_t i, int32_t tid, int orientation) {
    uint16_t f = rd->flag[i];
    int paired = (f & 0x1) != 0, mateMapped = !(f & 0x8);
    if (!paired || !mateMapped) return 0;
    if (tid != rd->mtid[i]) return 0;
    int rev = (f & 0x10) != 0, mrev = (f & 0x20) != 0;
    int diffStrand = rev != mrev;
    int posGap = !rev ? rd->pos[i] < rd->mpos[i] : rd->pos[i] > rd->mpos[i];
    if (orientation == ORC_OR_FR) return diffStrand && posGap;
    if (orientation == ORC_OR_RF) return diffStrand && !posGap;
    if (orientation == ORC_OR_FF) return !diffStrand && posGap;
    return 0;
}

static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return x < y ? -1 : x > y;
}

/* Junction::calcMetrics(orientation), junction.cc:683-687 */
static int calc_metrics(jsys *js, junc *J, int orientation) {
    const orc_reads *rd = js->rd;
    /* determineStrandFromReads :531-559 */
    uint32_t nb_pos = 0, nb_neg = 0, nb_unk = 0;
    for (size_t a = 0; a < J->n_al; a++) {
        switch (read_strand(rd, J->al[a].read)) {
        case ORC_STRAND_POS: nb_pos++; break;
        case ORC_STRAND_NEG: nb_neg++; break;
        default: nb_unk++; break;
        }
    }
    uint32_t total = nb_pos + nb_neg + nb_unk;
    const double threshold = 0.95;
    if ((double)nb_pos / (double)total >= threshold) J->r.read_strand = ORC_STRAND_POS;
    else if ((double)nb_neg / (double)total >= threshold) J->r.read_strand = ORC_STRAND_NEG;
    else J->r.read_strand = ORC_STRAND_UNK;
    /* calcEntropy :718-728 */
    int32_t *pp = (int32_t *)malloc((J->n_al ? J->n_al : 1) * sizeof(int32_t));
    if (!pp) return fail(ORC_ERR_NOMEM, "oom");
    for (size_t a = 0; a < J->n_al; a++) pp[a] = rd->pos[J->al[a].read];
    qsort(pp, J->n_al, sizeof(int32_t), cmp_i32);
    if (J->n_al > 1) J->r.entropy = orc_entropy(pp, J->n_al); /* n<=1 returns 0 without storing */
    free(pp);
    /* calcAlignmentStats :755-814 */
    int32_t lastStart = -1, lastEnd = -1;
    J->r.nb_dist = 0;
    J->r.nb_rel = 0;
    J->r.nb_up_juncs = 0;
    J->r.nb_down_juncs = 0;
    int properPairedCheck = orientation == ORC_OR_FR || orientation == ORC_OR_FF || orientation == ORC_OR_RF;
    for (size_t a = 0; a < J->n_al; a++) {
        int64_t i = J->al[a].read;
        int32_t start = rd->pos[i];
        int32_t end = rd->pos[i] + aligned_length(rd, i) - 1;
        if (start != lastStart || end != lastEnd) {
            J->r.nb_dist++;
            lastStart = start;
            lastEnd = end;
        }
        int reliable = 1;
        if (rd->mapq[i] >= 30) J->r.nb_um++; /* MAP_QUALITY_THRESHOLD junction.hpp:65 */
        else reliable = 0;
        if (rd->flag[i] & 0x2) J->r.nb_bpp++;
        if (properPairedCheck) {
            if (calc_if_proper_pair(rd, i, js->tid, orientation)) J->r.nb_ppp++;
            else reliable = 0;
        }
        if (reliable) J->r.nb_rel++;
        uint32_t upj = 0, downj = 0;
        int32_t pos = start;
        for (uint32_t k = rd->cig_off[i]; k < rd->cig_off[i + 1]; k++) {
            char type = op_chr(rd->cigar[k]);
            if (consumes_ref(type)) pos += op_len(rd->cigar[k]);
            if (type == 'N') {
                if (pos < J->r.start) upj++;
                else if (pos > J->r.end + 1) downj++;
            }
        }
        if (upj > J->r.nb_up_juncs) J->r.nb_up_juncs = upj;
        if (downj > J->r.nb_down_juncs) J->r.nb_down_juncs = downj;
    }
    return 0;
}

/* faidx_fetch_seq clamping, deps/htslib-1.3/faidx.c:439-476, on an in-memory contig. */
static void fetch_bases(const char *genome, int32_t glen, int32_t beg, int32_t end, sbuf *out) {
    if (end < beg) beg = end;
    if (beg < 0) beg = 0;
    else if (glen <= beg) beg = glen - 1;
    if (end < 0) end = 0;
    else if (glen <= end) end = glen - 1;
    out->n = 0;
    sb_reserve(out, 1);
    out->p[0] = 0;
    if (glen <= 0) return;
    sb_put(out, genome + beg, (size_t)(end - beg + 1));
}

static void to_upper(sbuf *s) {
    for (size_t i = 0; i < s->n; i++) s->p[i] = up(s->p[i]);
}

static const char NT16[] = "=ACMGRSVTWYHKDBN"; /* seq_nt16_str, deps/htslib-1.3/hts.c */

/* BamAlignment::getQuerySeq, bam_alignment.cc:244-250 */
static int get_query_seq(const orc_reads *rd, int64_t i, sbuf *out) {
    out->n = 0;
    if (sb_reserve(out, (size_t)(rd->l_qseq[i] > 0 ? rd->l_qseq[i] : 0) + 1)) return -1;
    out->p[0] = 0;
    const uint8_t *s = rd->seq4 + rd->seq_off[i];
    for (int32_t k = 0; k < rd->l_qseq[i]; k++) {
        uint8_t b = s[k >> 1];
        int c = (k & 1) ? (b & 0xf) : (b >> 4);
        out->p[out->n++] = NT16[c];
    }
    out->p[out->n] = 0;
    return 0;
}

/* AlignmentInfo::getNbMatchesFromStart / FromEnd, junction.cc:263-280 */
static uint32_t matches_from_start(const char *q, const char *a, size_t n) {
    for (size_t i = 0; i < n; i++)
        if (q[i] != a[i]) return (uint32_t)i;
    return (uint32_t)n;
}
static uint32_t matches_from_end(const char *q, const char *a, size_t n) {
    for (size_t j = n; j > 0; j--) {
        size_t i = j - 1;
        if (q[i] != a[i]) return (uint32_t)(n - i - 1);
    }
    return (uint32_t)n;
}

/* AlignmentInfo::calcMatchStats, junction.cc:147-240 */
static int calc_match_stats(jsys *js, junc *J, alninfo *A, const sbuf *ancLeft, const sbuf *ancRight,
                            sbuf *query, sbuf *qL, sbuf *qR, sbuf *gL, sbuf *gR) {
    const orc_reads *rd = js->rd;
    int64_t i = A->read;
    uint32_t leftStart = (uint32_t)J->r.left, rightEnd = (uint32_t)J->r.right;
    int32_t leftEnd = J->r.start - 1, rightStart = J->r.end + 1;
    int32_t qLeftStart = (int32_t)leftStart, qLeftEnd = leftEnd, qRightStart = rightStart, qRightEnd = (int32_t)rightEnd;
    if (get_query_seq(rd, i, query)) return fail(ORC_ERR_NOMEM, "oom");
    if (query->n <= 1) { /* :168-185 */
        A->totalUpMism = 0;
        A->totalDownMism = 0;
        A->totalUpMatches = (uint32_t)(leftEnd - (int32_t)leftStart + 1);
        A->totalDownMatches = (uint32_t)((int32_t)rightEnd - rightStart + 1);
        A->nbMismatches = 0;
        A->upMatches = 0;
        A->downMatches = 0;
        A->minMatch = 0;
        A->maxMatch = 0;
        A->mmes = A->totalUpMatches < A->totalDownMatches ? A->totalUpMatches : A->totalDownMatches;
        return 0;
    }
    const uint32_t *cig = rd->cigar + rd->cig_off[i];
    int nc = (int)(rd->cig_off[i + 1] - rd->cig_off[i]);
    int32_t pos = rd->pos[i], alen = aligned_length(rd, i);
    int rc;
    if ((rc = padded_query(cig, nc, pos, alen, query->p, query->n, (int32_t)leftStart, leftEnd, &qLeftStart, &qLeftEnd, qL))) return rc;
    if ((rc = padded_query(cig, nc, pos, alen, query->p, query->n, rightStart, (int32_t)rightEnd, &qRightStart, &qRightEnd, qR))) return rc;
    if ((rc = padded_genome(cig, nc, pos, alen, ancLeft->p, ancLeft->n, (int32_t)leftStart, leftEnd, qLeftStart, qLeftEnd, gL))) return rc;
    if ((rc = padded_genome(cig, nc, pos, alen, ancRight->p, ancRight->n, rightStart, (int32_t)rightEnd, qRightStart, qRightEnd, gR))) return rc;
    if (qL->n != gL->n || qL->n == 0)
        return fail(ORC_ERR_ANCHOR_MISMATCH, "Left anchor region for query and genome are not the same size (%zu vs %zu), read %lld", qL->n, gL->n, (long long)i);
    if (qR->n != gR->n || qR->n == 0)
        return fail(ORC_ERR_ANCHOR_MISMATCH, "Right Anchor region for query and genome are not the same size (%zu vs %zu), read %lld", qR->n, gR->n, (long long)i);
    int hl = orc_hamming(qL->p, qL->n, gL->p, gL->n);
    int hr = orc_hamming(qR->p, qR->n, gR->p, gR->n);
    A->totalUpMism = (uint32_t)hl;
    A->totalDownMism = (uint32_t)hr;
    A->totalUpMatches = (uint32_t)(qL->n - (size_t)hl);
    A->totalDownMatches = (uint32_t)(qR->n - (size_t)hr);
    A->nbMismatches = A->totalUpMism + A->totalDownMism;
    A->upMatches = matches_from_end(qL->p, gL->p, qL->n);
    A->downMatches = matches_from_start(qR->p, gR->p, qR->n);
    A->minMatch = A->upMatches < A->downMatches ? A->upMatches : A->downMatches;
    A->maxMatch = A->upMatches > A->downMatches ? A->upMatches : A->downMatches;
    A->mmes = A->totalUpMatches < A->totalDownMatches ? A->totalUpMatches : A->totalDownMatches;
    return 0;
}

/* Junction::processJunctionWindow, junction.cc:561-649 */
static int process_junction_window(jsys *js, junc *J) {
    sbuf donor = {0}, acceptor = {0}, leftAnc = {0}, rightAnc = {0}, leftInt = {0}, rightInt = {0};
    sbuf query = {0}, qL = {0}, qR = {0}, gL = {0}, gR = {0};
    int rc = 0;
    fetch_bases(js->genome, js->ref_len, J->r.start, J->r.start + 1, &donor);
    fetch_bases(js->genome, js->ref_len, J->r.end - 1, J->r.end, &acceptor);
    if (donor.n != 2 || acceptor.n != 2) {
        rc = fail(ORC_ERR_SPLICE_SITE_LEN, "Retrieved sequence for splice site of junction (%d,%d) is not the expected length", J->r.start, J->r.end);
        goto done;
    }
    to_upper(&donor);
    to_upper(&acceptor);
    {
        int ss, cons;
        int css = orc_donor_acceptor(donor.p, 2, acceptor.p, 2, J->r.read_strand, &ss, &cons, J->r.da1, J->r.da2);
        if (css < 0) { rc = css; goto done; }
        J->r.canonical = (uint8_t)css;
        J->r.ss_strand = (uint8_t)ss;
        J->r.cons_strand = (uint8_t)cons;
    }
    fetch_bases(js->genome, js->ref_len, J->r.left, J->r.start - 1, &leftAnc);
    fetch_bases(js->genome, js->ref_len, J->r.end + 1, J->r.right, &rightAnc);
    fetch_bases(js->genome, js->ref_len, J->r.start, J->r.start + 9, &leftInt);
    fetch_bases(js->genome, js->ref_len, J->r.end - 9, J->r.end, &rightInt);
    {
        int expLeftLen = J->r.start - J->r.left;
        if ((int)leftAnc.n != expLeftLen && expLeftLen > 0) { rc = fail(ORC_ERR_ANCHOR_LEN, "Retrieved sequence for left anchor of junction (%d,%d) is not the expected length", J->r.start, J->r.end); goto done; }
        int expRightLen = J->r.right - J->r.end;
        if ((int)rightAnc.n != expRightLen && expRightLen > 0) { rc = fail(ORC_ERR_ANCHOR_LEN, "Retrieved sequence for right anchor of junction (%d,%d) is not the expected length", J->r.start, J->r.end); goto done; }
        if (leftInt.n != 10 || rightInt.n != 10) { rc = fail(ORC_ERR_INTRON_FLANK_LEN, "Retrieved sequence for intron region of junction (%d,%d) is not the expected length", J->r.start, J->r.end); goto done; }
    }
    to_upper(&leftAnc);
    to_upper(&rightAnc);
    to_upper(&leftInt);
    to_upper(&rightInt);
    {
        const char *la10 = leftAnc.n < 10 ? leftAnc.p : leftAnc.p + (leftAnc.n - 10);
        size_t nla10 = leftAnc.n < 10 ? leftAnc.n : 10;
        size_t nra10 = rightAnc.n < 10 ? rightAnc.n : 10;
        rc = orc_hamming_scores(la10, nla10, leftInt.p, leftInt.n, rightInt.p, rightInt.n, rightAnc.p, nra10, J->r.cons_strand, &J->r.hamming5p, &J->r.hamming3p);
        if (rc) goto done;
    }
    for (size_t a = 0; a < J->n_al; a++) {
        rc = calc_match_stats(js, J, &J->al[a], &leftAnc, &rightAnc, &query, &qL, &qR, &gL, &gR);
        if (rc) goto done;
    }
    /* calcMismatchStats, junction.cc:862-909 (aJAD/junctionAnchorClarity never reach output) */
    {
        uint32_t nbMismatches = 0, firstMismatch = 100000000;
        for (size_t a = 0; a < J->n_al; a++) {
            alninfo *A = &J->al[a];
            if (A->mmes > J->r.maxmmes) J->r.maxmmes = A->mmes;
            nbMismatches += A->nbMismatches;
            if (A->minMatch > 0 && A->minMatch < firstMismatch) firstMismatch = A->minMatch;
            for (uint16_t k = 0; k < 20 && k < A->minMatch; k++) J->r.jad[k]++;
        }
        J->r.sum_mismatches = nbMismatches;
        J->r.mean_mismatches = (double)nbMismatches / (double)J->n_al;
        if (nbMismatches > 0 && firstMismatch < 20) {
            int found = 0;
            for (size_t a = 0; a < J->n_al; a++)
                if (J->al[a].minMatch > firstMismatch) { found = 1; break; }
            if (!found) J->r.suspicious = 1;
        }
    }
done:
    sb_free(&donor); sb_free(&acceptor); sb_free(&leftAnc); sb_free(&rightAnc);
    sb_free(&leftInt); sb_free(&rightInt); sb_free(&query); sb_free(&qL); sb_free(&qR);
    sb_free(&gL); sb_free(&gR);
    return rc;
}

/* JunctionBuilder::findJuncs, src/junction_builder.cc:314-357.  Junctions are
 * finalised after the read loop instead of as soon as `al.pos > intron.end`
 * (:324-331); with coordinate-sorted input the two are equivalent because no
 * later read can support an already-passed intron. */
int orc_find_juncs(int32_t tid, int32_t ref_len, const char *genome, const orc_reads *rd,
                   int orientation, orc_row **rows_out, int64_t *n_rows_out, orc_region *reg) {
    jsys js;
    memset(&js, 0, sizeof js);
    js.tid = tid;
    js.ref_len = ref_len;
    js.genome = genome;
    js.rd = rd;
    int rc = 0;
    uint64_t spliced = 0, unspliced = 0, sumLen = 0;
    int32_t minLen = INT32_MAX, maxLen = 0;
    *rows_out = NULL;
    *n_rows_out = 0;
    for (int64_t i = 0; i < rd->n; i++) {
        if (rd->xs[i] == 3) { rc = fail(ORC_ERR_BAD_XS, "Unknown strand (XS tag) on read %lld", (long long)i); goto done; }
        if (i > 0 && rd->pos[i] < rd->pos[i - 1]) { rc = fail(ORC_ERR_UNSORTED, "reads are not coordinate sorted at %lld", (long long)i); goto done; }
        int32_t len = rd->l_qseq[i];
        if (len < minLen) minLen = len;
        if (len > maxLen) maxLen = len;
        sumLen += (uint64_t)(int64_t)len;
        int found = 0;
        rc = add_junctions(&js, i, 0, rd->pos[i], &found);
        if (rc) goto done;
        if (found) spliced++;
        else unspliced++;
    }
    for (size_t j = 0; j < js.n; j++) {
        rc = calc_metrics(&js, &js.list[j], orientation);
        if (rc) goto done;
        rc = process_junction_window(&js, &js.list[j]);
        if (rc) goto done;
    }
    {
        orc_row *rows = (orc_row *)malloc((js.n ? js.n : 1) * sizeof(orc_row));
        if (!rows) { rc = fail(ORC_ERR_NOMEM, "oom"); goto done; }
        for (size_t j = 0; j < js.n; j++) rows[j] = js.list[j].r;
        *rows_out = rows;
        *n_rows_out = (int64_t)js.n;
    }
    if (reg) {
        reg->spliced = spliced;
        reg->unspliced = unspliced;
        reg->sum_len = sumLen;
        reg->min_len = minLen;
        reg->max_len = maxLen;
    }
done:
    for (size_t j = 0; j < js.n; j++) free(js.list[j].al);
    free(js.list);
    free(js.tab);
    return rc;
}

void orc_free_rows(orc_row *rows) { free(rows); }

static int cmp_row(const void *a, const void *b);
/* ------------------------------------------------------------------ */
/* junc --extra                                                       */
/* ------------------------------------------------------------------ */

/* libstdc++ std::_Hash_bytes, 64-bit (libstdc++-v3/libsupc++/hash_bytes.cc) */
static inline uint64_t hb_shift_mix(uint64_t v) { return v ^ (v >> 47); }
static uint64_t std_hash_bytes(const void *ptr, size_t len, uint64_t seed) {
    static const uint64_t mul = (((uint64_t)0xc6a4a793UL) << 32UL) + (uint64_t)0x5bd1e995UL;
    const unsigned char *buf = (const unsigned char *)ptr;
    const size_t len_aligned = len & ~(size_t)0x7;
    const unsigned char *end = buf + len_aligned;
    uint64_t hash = seed ^ (len * mul);
    for (const unsigned char *p = buf; p != end; p += 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        const uint64_t data = hb_shift_mix(w * mul) * mul;
        hash ^= data;
        hash *= mul;
    }
    if ((len & 0x7) != 0) {
        uint64_t data = 0;
        int n = (int)(len & 0x7);
        --n;
        do data = (data << 8) + end[n]; while (--n >= 0);
        hash ^= data;
        hash *= mul;
    }
    hash = hb_shift_mix(hash) * mul;
    hash = hb_shift_mix(hash);
    return hash;
}

/* BamAlignment::deriveName, bam_alignment.cc:233-242, then std::hash<string> (junction.hpp:158) */
uint64_t orc_name_hash(const char *qname, size_t n, uint16_t flag) {
    char tmp[300];
    if (n > 255) n = 255; /* l_read_name is 8 bits incl. NUL */
    memcpy(tmp, qname, n);
    size_t len = n;
    if (flag & 0x1) {
        const char *suf = (flag & 0x40) ? "_R1" : (flag & 0x80) ? "_R2" : "_R?";
        memcpy(tmp + len, suf, 3);
        len += 3;
    }
    return std_hash_bytes(tmp, len, 0xc70f6907UL);
}

static int has_refskip(const orc_reads *rd, int64_t i) { /* BamAlignment::isSplicedRead, bam_alignment.cc:294-301 */
    for (uint32_t k = rd->cig_off[i]; k < rd->cig_off[i + 1]; k++)
        if (op_chr(rd->cigar[k]) == 'N') return 1;
    return 0;
}
/* a record of unspliced.bam: not spliced, and mapped (src/junction_builder.cc:168-186) */
static int is_unspliced_mapped(const orc_reads *rd, int64_t i) { return !has_refskip(rd, i) && !(rd->flag[i] & 0x4); }

/* bam_endpos (deps/htslib-1.3/sam.c: bam_cigar2rlen, 0 -> pos + 1) */
static int32_t end_excl(const orc_reads *rd, int64_t i) {
    int32_t a = aligned_length(rd, i);
    return rd->pos[i] + (a > 0 ? a : 1);
}

/* The pileup of unspliced.bam for one target, deps/htslib-1.3/sam.c:1853-1975 driven by
 * DepthParser::read_bam_skip_gapped (depth_parser.cc:60-83; min_mapQ 0 and min_len 0 filter nothing).
 * bam_plp_push keeps every mapped record except: one that starts at the position the iterator stands on
 * (the start of the last record it kept) while more than maxcnt = 8000 list nodes are allocated -- the
 * records not yet passed (end > every position already piled up) plus the list's sentinel and the dummy.
 * Every position covered by a kept record is reported; its depth is the number of kept records there
 * minus those in a deletion (is_del; is_refskip cannot occur without N operations). */
int64_t orc_depth(int32_t ref_len, const orc_reads *rd, uint32_t *depth) {
    if (ref_len < 0) return fail(ORC_ERR_ARG, "bad ref_len");
    memset(depth, 0, (size_t)ref_len * sizeof(uint32_t));
    /* ends of the kept records still in the list, as a multiset by value: cnt_end[e] */
    uint32_t *cnt_end = (uint32_t *)calloc((size_t)ref_len + 2, sizeof(uint32_t));
    int32_t *diff = (int32_t *)calloc((size_t)ref_len + 2, sizeof(int32_t));
    if (!cnt_end || !diff) { free(cnt_end); free(diff); return fail(ORC_ERR_NOMEM, "oom"); }
    int64_t kept = 0, in_list = 0;
    int32_t iter_pos = -1; /* start of the last kept record of this target (none yet) */
    int32_t swept = 0;     /* ends < swept have been taken out of in_list */
    for (int64_t i = 0; i < rd->n; i++) {
        if (!is_unspliced_mapped(rd, i)) continue;
        const int32_t p = rd->pos[i];
        int32_t e = end_excl(rd, i);
        if (p < 0) continue;
        /* the iterator has piled up every position < iter_pos: records with end <= iter_pos - 1 ... are freed
         * when position (end) is visited, i.e. all ends <= iter_pos - 1 + 1 - 1; the list therefore holds the
         * kept records with end >= iter_pos (sam.c:1861-1864 frees `end <= pos` at each visited pos < iter_pos) */
        if (iter_pos >= 0) {
            while (swept < iter_pos && swept <= ref_len + 1) { in_list -= cnt_end[swept]; swept++; }
        }
        if (iter_pos == p && in_list + 2 > 8000) continue; /* sam.c:1906-1910 */
        kept++;
        in_list++;
        { int32_t ce = e > ref_len + 1 ? ref_len + 1 : e; if (ce < swept) ce = swept; cnt_end[ce]++; }
        iter_pos = p;
        /* depth contribution: M / = / X runs */
        int32_t x = p;
        for (uint32_t k = rd->cig_off[i]; k < rd->cig_off[i + 1]; k++) {
            char t = op_chr(rd->cigar[k]);
            int32_t l = op_len(rd->cigar[k]);
            if (t == 'M' || t == '=' || t == 'X') {
                int32_t a = x, b = x + l;
                if (a < 0) a = 0;
                if (b > ref_len) b = ref_len;
                if (a < b) { diff[a]++; diff[b]--; }
            }
            if (consumes_ref(t)) x += l;
        }
    }
    int64_t run = 0;
    for (int32_t x = 0; x < ref_len; x++) {
        run += diff[x];
        if (x + 1 < ref_len) depth[x + 1] = (uint32_t)run; /* depths[pos + 1] = cnt, depth_parser.cc:127,147-150 */
    }
    free(cnt_end);
    free(diff);
    return kept;
}

/* Junction::calcCoverage(int32_t a, int32_t b, const vector<uint32_t>&), junction.cc:923-933 */
static double cov_window(int32_t a, int32_t b, const uint32_t *lv, size_t n) {
    double multiplier = 1.0 / (b - a);
    uint32_t readCount = 0;
    for (int32_t i = a; i <= b; i++)
        if (i >= 0 && i < (int32_t)n) readCount += lv[i];
    return multiplier * (double)readCount;
}
double orc_calc_coverage(int32_t start, int32_t end, const uint32_t *lv, size_t n) { /* junction.cc:935-951 */
    const int32_t REGION_LENGTH = 10;
    int32_t donorStart = start - 2 * REGION_LENGTH, donorMid = start - REGION_LENGTH, donorEnd = start;
    int32_t acceptorStart = end, acceptorMid = end + REGION_LENGTH, acceptorEnd = end + 2 * REGION_LENGTH;
    double donorCoverage = cov_window(donorStart, donorMid - 1, lv, n) - cov_window(donorMid, donorEnd, lv, n);
    double acceptorCoverage = cov_window(acceptorMid, acceptorEnd, lv, n) - cov_window(acceptorStart, acceptorMid - 1, lv, n);
    return donorCoverage + acceptorCoverage;
}

typedef struct { uint64_t code; uint32_t count; } nslot;

int orc_extra(int32_t n_refs, const int32_t *ref_len, const orc_reads *reads, const uint64_t *const *name_hash,
              orc_row *rows, int64_t n_rows, int32_t max_query_len) {
    int rc = 0;
    /* ---- splicedAlignmentMap[hash(deriveName)]++ for every spliced record of the file, junction_builder.cc:168-176 */
    size_t n_spliced = 0;
    for (int32_t t = 0; t < n_refs; t++)
        for (int64_t i = 0; i < reads[t].n; i++) n_spliced += (size_t)has_refskip(&reads[t], i);
    size_t cap = 16;
    while (cap < 2 * n_spliced + 1) cap <<= 1;
    nslot *map = (nslot *)calloc(cap, sizeof(nslot));
    if (!map) return fail(ORC_ERR_NOMEM, "oom");
    for (int32_t t = 0; t < n_refs; t++)
        for (int64_t i = 0; i < reads[t].n; i++) {
            if (!has_refskip(&reads[t], i)) continue;
            uint64_t c = name_hash[t][i];
            size_t h = (size_t)(c * 0x9e3779b97f4a7c15ULL) & (cap - 1);
            while (map[h].count && map[h].code != c) h = (h + 1) & (cap - 1);
            map[h].code = c;
            map[h].count++;
        }
    /* ---- per target: group again (JunctionSystem::addJunctions) to get every junction's alignment list */
    int64_t r0 = 0;
    for (int32_t t = 0; t < n_refs && !rc; t++) {
        int64_t r1 = r0;
        while (r1 < n_rows && rows[r1].refid == t) r1++;
        const orc_reads *rd = &reads[t];
        if (r1 > r0) {
            jsys js;
            memset(&js, 0, sizeof js);
            js.tid = t;
            js.ref_len = ref_len[t];
            js.rd = rd;
            for (int64_t i = 0; i < rd->n && !rc; i++) {
                int found = 0;
                rc = add_junctions(&js, i, 0, rd->pos[i], &found);
            }
            /* calcMultipleMappingScore, junction.cc:914-921: N / M, M a uint32_t sum of map[code] */
            for (size_t j = 0; j < js.n && !rc; j++) {
                junc *J = &js.list[j];
                orc_row key = J->r, *row;
                key.refid = t;
                row = (orc_row *)bsearch(&key, rows + r0, (size_t)(r1 - r0), sizeof(orc_row), cmp_row);
                if (!row) { rc = fail(ORC_ERR_ARG, "orc_extra: rows do not match the reads"); break; }
                uint32_t M = 0;
                for (size_t a = 0; a < J->n_al; a++) {
                    uint64_t c = name_hash[t][J->al[a].read];
                    size_t h = (size_t)(c * 0x9e3779b97f4a7c15ULL) & (cap - 1);
                    while (map[h].count && map[h].code != c) h = (h + 1) & (cap - 1);
                    M += map[h].count;
                }
                row->mm_score = (double)J->n_al / (double)M;
            }
            for (size_t j = 0; j < js.n; j++) free(js.list[j].al);
            free(js.list);
            free(js.tab);
        }
        /* ---- Junction::processJunctionVicinity over unspliced.bam, junction.cc:651-677.  The region query
         * (sam_itr_queryi: records with pos < regionEnd and bam_endpos > regionStart) never cuts off a record
         * the two tests accept, but it is applied as written. */
        int32_t maxspan = 1; /* longest reference span of a record: nothing further left than this can reach a region */
        for (int64_t i = 0; i < rd->n && r1 > r0; i++) {
            int32_t a = aligned_length(rd, i);
            if (a > maxspan) maxspan = a;
        }
        for (int64_t r = r0; r < r1 && !rc; r++) {
            orc_row *row = &rows[r];
            int32_t regionStart = row->left - max_query_len - 1;
            regionStart = regionStart < 0 ? 0 : regionStart;
            int32_t regionEnd = row->right + max_query_len + 1;
            regionEnd = regionEnd >= ref_len[t] ? ref_len[t] - 1 : regionEnd;
            uint32_t up = 0, down = 0;
            /* records are sorted by pos: stop at pos >= regionEnd */
            int64_t lo = 0, hi = rd->n; /* first record with pos > regionStart - maxspan */
            while (lo < hi) {
                int64_t mid = (lo + hi) / 2;
                if ((int64_t)rd->pos[mid] + maxspan <= regionStart) lo = mid + 1;
                else hi = mid;
            }
            for (int64_t i = lo; i < rd->n && rd->pos[i] < regionEnd; i++) {
                if (!is_unspliced_mapped(rd, i)) continue;
                if (end_excl(rd, i) <= regionStart) continue;
                int32_t pos = rd->pos[i], end = pos + aligned_length(rd, i) - 1; /* getStart / getEnd */
                if (row->start > pos && row->left <= end) up++;
                if (row->right >= pos && row->end < pos) down++;
            }
            row->up_aln = up;
            row->down_aln = down;
        }
        r0 = r1;
    }
    /* ---- JunctionSystem::calcCoverage, junction_system.cc:231-242.  loadNextBatch (depth_parser.cc:112-164)
     * fills the vector of the target the previous call stopped in and stops at the first pileup position of
     * the NEXT target, which it records in `last`; getCurrentRefIndex() then names that next target, so a
     * batch is applied to the junctions of the target AFTER the one it was computed for -- except the last
     * batch (the pileup ended: `last` still names its own target).  Targets without an unspliced record never
     * appear. */
    if (!rc) {
        int32_t prev = -1;
        uint32_t *depth_prev = NULL;
        int32_t last = -1;
        for (int32_t t = 0; t < n_refs; t++) {
            int any = 0;
            for (int64_t i = 0; i < reads[t].n && !any; i++) any = is_unspliced_mapped(&reads[t], i);
            if (any) last = t;
        }
        for (int32_t t = 0; t < n_refs && !rc; t++) {
            int any = 0;
            for (int64_t i = 0; i < reads[t].n && !any; i++) any = is_unspliced_mapped(&reads[t], i);
            if (!any) continue;
            uint32_t *depth = (uint32_t *)malloc(((size_t)ref_len[t] + 1) * sizeof(uint32_t));
            if (!depth) { rc = fail(ORC_ERR_NOMEM, "oom"); break; }
            if (orc_depth(ref_len[t], &reads[t], depth) < 0) { free(depth); rc = ORC_ERR_NOMEM; break; }
            if (prev >= 0) /* the batch of `prev` is handed to the junctions of `t` */
                for (int64_t r = 0; r < n_rows; r++)
                    if (rows[r].refid == t)
                        rows[r].coverage = orc_calc_coverage(rows[r].start, rows[r].end, depth_prev, (size_t)ref_len[prev]);
            if (t == last)
                for (int64_t r = 0; r < n_rows; r++)
                    if (rows[r].refid == t)
                        rows[r].coverage = orc_calc_coverage(rows[r].start, rows[r].end, depth, (size_t)ref_len[t]);
            free(depth_prev);
            depth_prev = depth;
            prev = t;
        }
        free(depth_prev);
    }
    free(map);
    return rc;
}

/* ------------------------------------------------------------------ */
/* merge: sort / index / calcJunctionStats                            */
/* ------------------------------------------------------------------ */

/* JunctionComparator -> IntronComparator, junction.hpp:1415-1420, intron.cc:111-127 */
static int cmp_row(const void *a, const void *b) {
    const orc_row *x = (const orc_row *)a, *y = (const orc_row *)b;
    if (x->refid != y->refid) return x->refid < y->refid ? -1 : 1;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    if (x->end != y->end) return x->end < y->end ? -1 : 1;
    return 0;
}

static int shares(const orc_row *a, const orc_row *b) { /* intron.cc:55-58 */
    return a->refid == b->refid && (a->start == b->start || a->end == b->end);
}

void orc_finalize(orc_row *rows, int64_t n, double meanQueryLength) {
    qsort(rows, (size_t)n, sizeof(orc_row), cmp_row); /* junction_system.cc:322-324 */
    for (int64_t i = 0; i < n; i++) rows[i].id = (uint32_t)i; /* :326-330 */
    if (n <= 1) return; /* src/junction_builder.cc:285 */
    /* calcJunctionStats, junction_system.cc:250-320; createJunctionGroup :55-70 */
    for (int64_t i = 0; i < n; i++) {
        int64_t gs = i, ge = i; /* group = rows[gs..ge] */
        {
            int64_t cur = i, ret = n - 1;
            for (int64_t j = i + 1; j < n; j++) {
                if (shares(&rows[cur], &rows[j])) { ge = j; cur = j; }
                else { ret = j - 1; break; }
            }
            i = ret;
        }
        uint32_t maxReads = 0;
        int64_t maxIndex = 0;
        int uniqueJunction = (ge - gs + 1) == 1;
        for (int64_t j = 0; j <= ge - gs; j++) {
            orc_row *r = &rows[gs + j];
            if (maxReads < r->nb_raw) { maxReads = r->nb_raw; maxIndex = j; }
            r->uniq = (uint8_t)uniqueJunction;
        }
        rows[gs + maxIndex].primary = 1;
    }
    {
        int64_t i = 0;
        int lastdiffseq = 0;
        while (i < n - 1) {
            orc_row *first = &rows[i], *second = &rows[i + 1];
            int32_t diff = second->start - first->end;
            diff = diff < 0 ? 0 : diff;
            if (first->refid != second->refid) {
                first->dist_up = (uint32_t)-1;
                second->dist_down = (uint32_t)-1;
                if (i == 0 || lastdiffseq) first->dist_down = (uint32_t)-1;
                if (i == n - 2) second->dist_up = (uint32_t)-1;
                lastdiffseq = 1;
            } else if (i == 0) {
                first->dist_down = (uint32_t)-1;
                first->dist_up = (uint32_t)diff;
                second->dist_down = (uint32_t)diff;
                lastdiffseq = 0;
            } else if (i == n - 2) {
                first->dist_up = (uint32_t)diff;
                second->dist_down = (uint32_t)diff;
                second->dist_up = (uint32_t)-1;
                lastdiffseq = 0;
            } else {
                first->dist_up = (uint32_t)diff;
                second->dist_down = (uint32_t)diff;
                lastdiffseq = 0;
            }
            i++;
        }
    }
    for (int64_t i = 0; i < n; i++) {
        orc_row *r = &rows[i];
        int32_t down = (int32_t)r->dist_down, upd = (int32_t)r->dist_up;
        int32_t nearest = (down == -1 || upd == -1) ? (down > upd ? down : upd) : (down < upd ? down : upd);
        r->dist_nearest = (uint32_t)nearest;
        r->mean_readlen = (double)(uint32_t)meanQueryLength; /* setMeanReadLength(uint32_t), junction.hpp:928 */
        if (r->suspicious) {
            double prob = 1.0 - pow(((double)r->maxmmes / (meanQueryLength / 2.0)), (double)r->nb_raw);
            if (prob > 0.99) r->pfp = 1;
        }
    }
}

/* JunctionSystem::determineStrandedness, junction_system.cc:455-560 */
void orc_determine_strandedness(const orc_row *rows, int64_t n, int *orientation, int *strandedness) {
    uint32_t pp1 = 0, pn1 = 0, pp2 = 0, pn2 = 0, np1 = 0, nn1 = 0, np2 = 0, nn2 = 0; /* tot_r{1,2}_{pos,neg}_when_ss_{pos,neg} */
    for (int64_t i = 0; i < n; i++) {
        const orc_row *j = &rows[i];
        if (j->ss_strand == ORC_STRAND_POS) { pp1 += j->r1pos; pn1 += j->r1neg; pp2 += j->r2pos; pn2 += j->r2neg; }
        else if (j->ss_strand == ORC_STRAND_NEG) { np1 += j->r1pos; nn1 += j->r1neg; np2 += j->r2pos; nn2 += j->r2neg; }
    }
    double posr1 = ((double)((int32_t)pp1 - (int32_t)pn1)) / ((double)(pp1 + pn1));
    double negr1 = ((double)((int32_t)nn1 - (int32_t)np1)) / ((double)(np1 + nn1));
    double posr2 = ((double)((int32_t)pp2 - (int32_t)pn2)) / ((double)(pp2 + pn2));
    double negr2 = ((double)((int32_t)nn2 - (int32_t)np2)) / ((double)(np2 + nn2));
    uint32_t totalr1 = pp1 + pn1 + np1 + nn1, totalr2 = pp2 + pn2 + np2 + nn2;
    int s = 3 /* UNKNOWN */, o = ORC_OR_UNKNOWN;
    if (totalr1 == 0 && totalr2 == 0) {
    } else if (totalr2 == 0) {
        o = ORC_OR_SE;
        if (posr1 > 0.5 && negr1 > 0.5) s = 2;
        else if (posr1 < -0.5 && negr1 < -0.5) s = 1;
    } else {
        o = ORC_OR_FR;
        if (posr1 > 0.5 && negr1 > 0.5 && posr2 < -0.5 && negr2 < -0.5) s = 2;
        else if (posr1 < -0.5 && negr1 < -0.5 && posr2 > 0.5 && negr2 > 0.5) s = 1;
        else if (posr1 > 0.5 && negr1 > 0.5 && posr2 > 0.5 && negr2 > 0.5) { s = 2; o = ORC_OR_FF; }
        else if (posr1 < -0.5 && negr1 < -0.5 && posr2 < -0.5 && negr2 < -0.5) { s = 1; o = ORC_OR_FF; }
    }
    if (fabs(posr1) <= 0.5 && fabs(negr1) <= 0.5 && fabs(posr2) <= 0.5 && fabs(negr2) <= 0.5) s = 0;
    *orientation = o;
    *strandedness = s;
}

/* ------------------------------------------------------------------ */
/* bamfilt                                                            */
/* ------------------------------------------------------------------ */
static int js_has(const int32_t *S, const int32_t *E, int64_t n, int32_t s, int32_t e) { /* JunctionSystem::getJunction */
    for (int64_t i = 0; i < n; i++)
        if (S[i] == s && E[i] == e) return 1;
    return 0;
}
/* BamFilter::containsJunctionInSystem, src/bam_filter.cc:75-100 */
static int contains_junction(const orc_reads *rd, int64_t i, const int32_t *S, const int32_t *E, int64_t n) {
    int32_t lEnd = rd->pos[i], rStart;
    for (uint32_t k = rd->cig_off[i]; k < rd->cig_off[i + 1]; k++) {
        char t = op_chr(rd->cigar[k]);
        int32_t len = op_len(rd->cigar[k]);
        if (t == 'N') {
            rStart = lEnd + len;
            if (js_has(S, E, n, lEnd, rStart - 1)) return 1;
        } else if (consumes_ref(t)) {
            lEnd += len;
        }
    }
    return 0;
}
/* BamFilter::clipMSR, src/bam_filter.cc:102-150: only `allBad` reaches the output */
static int clip_msr_all_bad(const orc_reads *rd, int64_t i, const int32_t *S, const int32_t *E, int64_t n) {
    int32_t lEnd = rd->pos[i], rStart;
    int ab = 1;
    for (uint32_t k = rd->cig_off[i]; k < rd->cig_off[i + 1]; k++) {
        char t = op_chr(rd->cigar[k]);
        int32_t len = op_len(rd->cigar[k]);
        if (t == 'N') {
            rStart = lEnd + len;
            if (js_has(S, E, n, lEnd, rStart - 1)) ab = 0;
        } else if (consumes_ref(t)) {
            lEnd += len;
        }
    }
    return ab;
}
int orc_bamfilt_flags(const orc_reads *rd, const int32_t *S, const int32_t *E, int64_t n_js, int clip_mode, uint8_t *out) {
    for (int64_t i = 0; i < rd->n; i++) {
        if (has_refskip(rd, i)) {
            if (clip_mode == 2 || nb_junctions_in_read(rd, i) <= 1) out[i] = contains_junction(rd, i, S, E, n_js) ? 2 : 0;
            else out[i] = clip_msr_all_bad(rd, i, S, E, n_js) ? 0 : 3;
        } else
            out[i] = 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* filt: Markov models and the feature rows                           */
/* ------------------------------------------------------------------ */
/* SeqUtils::makeClean, seq_utils.hpp:54-60: upper-case, anything but A C G T becomes N */
static int clean_code(char c) {
    c = up(c);
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
}
static void make_clean(const sbuf *in, sbuf *out) {
    out->n = 0;
    for (size_t i = 0; i < in->n; i++) {
        char c = "ACGTN"[clean_code(in->p[i])];
        sb_put(out, &c, 1);
    }
}
/* KmerMarkovModel (lib/src/markov_model.cc:31-78) as a dense table over the cleaned alphabet; `seen` marks the
 * contexts the unordered_map holds (model.size() of :88-90 is their count) */
typedef struct {
    int order;
    double *tab;      /* [5^order][5] counts, then probabilities */
    uint8_t *seen;    /* [5^order] */
    size_t n_ctx;
} kmm;
static size_t pow5(int k) { size_t r = 1; while (k-- > 0) r *= 5; return r; }
static int kmm_init(kmm *m, int order) {
    m->order = order;
    m->n_ctx = pow5(order);
    m->tab = (double *)calloc(m->n_ctx * 5, sizeof(double));
    m->seen = (uint8_t *)calloc(m->n_ctx, 1);
    return m->tab && m->seen ? 0 : -1;
}
static void kmm_free(kmm *m) { free(m->tab); free(m->seen); }
static size_t ctx_index(const char *s, int order) {
    size_t x = 0;
    for (int k = 0; k < order; k++) x = x * 5 + (size_t)clean_code(s[k]);
    return x;
}
static void kmm_count(kmm *m, const sbuf *clean) { /* train, :34-41: only strings longer than order + 1 */
    if (clean->n > (size_t)m->order + 1)
        for (size_t i = (size_t)m->order; i < clean->n; i++) {
            size_t c = ctx_index(clean->p + i - m->order, m->order);
            m->tab[c * 5 + (size_t)clean_code(clean->p[i])] += 1.0;
            m->seen[c] = 1;
        }
}
static void kmm_normalise(kmm *m) { /* :43-53 */
    for (size_t c = 0; c < m->n_ctx; c++) {
        double sum = 0;
        for (int k = 0; k < 5; k++) sum += m->tab[c * 5 + k];
        if (sum > 0)
            for (int k = 0; k < 5; k++) m->tab[c * 5 + k] = m->tab[c * 5 + k] / sum;
    }
}
static size_t kmm_size(const kmm *m) { size_t n = 0; for (size_t c = 0; c < m->n_ctx; c++) n += m->seen[c]; return n; }
static double kmm_score(kmm *m, const sbuf *raw) { /* getScore, :57-78 (operator[] inserts the contexts it looks up) */
    sbuf s = {0};
    make_clean(raw, &s);
    double score = 1.0;
    uint32_t no_count = 0;
    for (size_t i = (size_t)m->order; i < s.n; i++) {
        size_t c = ctx_index(s.p + i - m->order, m->order);
        m->seen[c] = 1;
        double v = m->tab[c * 5 + (size_t)clean_code(s.p[i])];
        if (v != 0.0) score *= v;
        else no_count++;
    }
    sb_free(&s);
    if (score == 0.0) return -100.0;
    else if (no_count > 2) score /= ((double)no_count * 0.5);
    return log(score);
}
/* PosMarkovModel, markov_model.cc:80-115 */
typedef struct { int order; double tab[ORC_PW_LEN][5]; uint8_t seen[ORC_PW_LEN]; } pmm;
static void pmm_count(pmm *m, const sbuf *clean) {
    for (size_t i = (size_t)m->order; i < clean->n && i < ORC_PW_LEN; i++) {
        m->tab[i][clean_code(clean->p[i])] += 1.0;
        m->seen[i] = 1;
    }
}
static void pmm_normalise(pmm *m) {
    for (int i = 0; i < ORC_PW_LEN; i++) {
        double sum = 0;
        for (int k = 0; k < 5; k++) sum += m->tab[i][k];
        if (sum > 0)
            for (int k = 0; k < 5; k++) m->tab[i][k] = m->tab[i][k] / sum;
    }
}
static size_t pmm_size(const pmm *m) { size_t n = 0; for (int i = 0; i < ORC_PW_LEN; i++) n += m->seen[i]; return n; }
static double pmm_score(pmm *m, const sbuf *raw) {
    sbuf s = {0};
    make_clean(raw, &s);
    double score = 1.0;
    for (size_t i = (size_t)m->order; i < s.n && i < ORC_PW_LEN; i++) {
        m->seen[i] = 1;
        score *= m->tab[i][clean_code(s.p[i])];
    }
    sb_free(&s);
    if (score == 0.0) return -300.0;
    return log(score);
}
/* gmap.fetchBases + SeqUtils::reverseComplement when the consensus strand is negative.  The reference does not
 * upper-case the fetched bases first, so a lower-case letter indexes REVCOMP_LOOKUP (26 entries from 'A') out of
 * bounds there -- undefined behaviour.  Here the bases are upper-cased first, as the junc stage does with everything it
 * fetches (junction.cc:586-587,635-638); for upper-case contigs the two are the same thing. */
static void fetch_oriented(const char *genome, int32_t glen, int32_t beg, int32_t end, int neg, sbuf *out) {
    out->n = 0;
    fetch_bases(genome, glen, beg, end, out);
    to_upper(out);
    if (neg && out->n) {
        char *tmp = (char *)malloc(out->n);
        orc_revcomp(out->p, out->n, tmp);
        memcpy(out->p, tmp, out->n);
        free(tmp);
    }
}
static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

int orc_filt_features(int32_t n_refs, const int32_t *ref_len, const char *const *genomes, const orc_row *rows, int64_t n_rows,
                      const int64_t *l95_idx, int64_t n_l95, const int64_t *cp_idx, int64_t n_cp, const int64_t *pass_idx,
                      int64_t n_pass, const int64_t *fail_idx, int64_t n_fail, double *F, double *models_out, uint32_t *l95_out) {
    kmm exon, intron, donT, donF, accT, accF;
    pmm donP, accP;
    memset(&donP, 0, sizeof donP);
    memset(&accP, 0, sizeof accP);
    donP.order = accP.order = 1;
    if (kmm_init(&exon, 5) || kmm_init(&intron, 5) || kmm_init(&donT, 5) || kmm_init(&donF, 5) || kmm_init(&accT, 5) || kmm_init(&accF, 5))
        return fail(ORC_ERR_NOMEM, "oom");
    sbuf a = {0}, b = {0}, cl = {0};
    /* calcIntronThreshold, model_features.cc:67-75 */
    uint32_t L95 = 0;
    if (n_l95 > 0) {
        uint32_t *sz = (uint32_t *)malloc((size_t)n_l95 * 4);
        for (int64_t k = 0; k < n_l95; k++) sz[k] = (uint32_t)(rows[l95_idx[k]].end - rows[l95_idx[k]].start + 1);
        qsort(sz, (size_t)n_l95, 4, cmp_u32);
        L95 = sz[(size_t)((double)n_l95 * 0.95)];
        free(sz);
    }
    if (l95_out) *l95_out = L95;
    /* trainCodingPotentialModel, model_features.cc:77-112 */
    for (int64_t k = 0; k < n_cp; k++) {
        const orc_row *j = &rows[cp_idx[k]];
        const char *g = genomes[j->refid];
        const int32_t gl = ref_len[j->refid];
        const int neg = j->cons_strand == ORC_STRAND_NEG;
        fetch_oriented(g, gl, j->start - 202, j->start - 2, neg, &a); make_clean(&a, &cl); kmm_count(&exon, &cl);
        fetch_oriented(g, gl, j->start, j->end, neg, &a);             make_clean(&a, &cl); kmm_count(&intron, &cl);
        fetch_oriented(g, gl, j->end + 1, j->end + 201, neg, &a);     make_clean(&a, &cl); kmm_count(&exon, &cl);
    }
    kmm_normalise(&exon);
    kmm_normalise(&intron);
    /* trainSplicingModels, model_features.cc:114-158 */
    for (int pass = 0; pass < 2; pass++) {
        const int64_t *idx = pass == 0 ? pass_idx : fail_idx;
        const int64_t n = pass == 0 ? n_pass : n_fail;
        for (int64_t k = 0; k < n; k++) {
            const orc_row *j = &rows[idx[k]];
            const char *g = genomes[j->refid];
            const int32_t gl = ref_len[j->refid];
            const int neg = j->cons_strand == ORC_STRAND_NEG;
            fetch_oriented(g, gl, j->start - 3, j->start + 20, neg, &a); /* left */
            fetch_oriented(g, gl, j->end - 20, j->end + 2, neg, &b);     /* right */
            const sbuf *don = neg ? &b : &a, *acc = neg ? &a : &b;
            make_clean(don, &cl);
            if (pass == 0) { pmm_count(&donP, &cl); kmm_count(&donT, &cl); } else kmm_count(&donF, &cl);
            make_clean(acc, &cl);
            if (pass == 0) { pmm_count(&accP, &cl); kmm_count(&accT, &cl); } else kmm_count(&accF, &cl);
        }
    }
    pmm_normalise(&donP); pmm_normalise(&accP);
    kmm_normalise(&donT); kmm_normalise(&accT); kmm_normalise(&donF); kmm_normalise(&accF);
    if (models_out) {
        double *o = models_out;
        const kmm *ks[6] = {&exon, &intron, &donT, &donF, &accT, &accF};
        for (int m = 0; m < 6; m++) { memcpy(o, ks[m]->tab, ORC_KMER_TABLE * sizeof(double)); o += ORC_KMER_TABLE; }
        memcpy(o, donP.tab, sizeof donP.tab); o += ORC_PW_LEN * 5;
        memcpy(o, accP.tab, sizeof accP.tab); o += ORC_PW_LEN * 5;
        o[0] = (double)kmm_size(&exon); o[1] = (double)kmm_size(&intron); o[2] = (double)pmm_size(&donP); o[3] = (double)pmm_size(&accP);
        o[4] = o[5] = o[6] = o[7] = 0;
    }
    /* setRow, model_features.cc:161-212 */
    for (int64_t r = 0; r < n_rows; r++) {
        const orc_row *j = &rows[r];
        const char *g = genomes[j->refid];
        const int32_t gl = ref_len[j->refid];
        const int neg = j->cons_strand == ORC_STRAND_NEG;
        double *f = F + (size_t)r * ORC_N_FEATURES;
        /* calcSplicingScores, junction.cc:1361-1382 (always evaluated, and its lookups insert into the maps) */
        fetch_oriented(g, gl, j->start - 3, j->start + 20, neg, &a);
        fetch_oriented(g, gl, j->end - 20, j->end + 2, neg, &b);
        const sbuf *don = neg ? &b : &a, *acc = neg ? &a : &b;
        const double pws = pmm_score(&donP, don) + pmm_score(&accP, acc);
        const double ss = (kmm_score(&donT, don) - kmm_score(&donF, don)) + (kmm_score(&accT, acc) - kmm_score(&accF, acc));
        const uint32_t size = (uint32_t)(j->end - j->start + 1);
        f[0] = 0.0; /* isGenuine(): false unless a truth set marked it */
        f[1] = (double)(j->nb_raw - j->nb_ms);
        f[2] = (double)j->nb_dist;
        f[3] = (double)j->nb_rel;
        f[4] = j->entropy;
        f[5] = (double)j->nb_rel / (double)j->nb_raw;
        f[6] = (double)j->max_min_anc;
        f[7] = (double)j->maxmmes;
        f[8] = j->mean_mismatches;
        f[9] = L95 == 0 ? 0.0 : (size <= L95 ? 0.0 : log((double)(size - L95))); /* calcIntronScore, junction.cc:953-956 */
        f[10] = (double)(j->hamming5p < j->hamming3p ? j->hamming5p : j->hamming3p);
        if (kmm_size(&exon) == 0 || kmm_size(&intron) == 0) f[11] = 0.0;
        else { /* calcCodingPotential, junction.cc:1328-1359 */
            double cp = 0;
            fetch_oriented(g, gl, j->start - 82, j->start - 2, neg, &a);
            cp = (kmm_score(&exon, &a) - kmm_score(&intron, &a));
            fetch_oriented(g, gl, j->start, j->start + 80, neg, &a);
            cp = cp + (kmm_score(&intron, &a) - kmm_score(&exon, &a));
            fetch_oriented(g, gl, j->end - 80, j->end, neg, &a);
            cp = cp + (kmm_score(&intron, &a) - kmm_score(&exon, &a));
            fetch_oriented(g, gl, j->end + 1, j->end + 81, neg, &a);
            cp = cp + (kmm_score(&exon, &a) - kmm_score(&intron, &a));
            f[11] = cp;
        }
        const int pw_empty = pmm_size(&donP) == 0 || pmm_size(&accP) == 0;
        f[12] = pw_empty ? 0.0 : pws;
        f[13] = pw_empty ? 0.0 : ss;
        for (int i = 0; i < 20; i++) { /* calcJunctionAnchorDepthLogDeviation, junction.cc:1384-1391 */
            double Ni = (double)j->jad[i];
            if (Ni == 0.0) Ni = 0.000000000001;
            double Pi = 1.0 - ((double)i / (double)(j->mean_readlen / 2.0));
            double Ei = (double)j->nb_raw * Pi;
            f[14 + i] = log2(Ni / Ei);
        }
    }
    sb_free(&a); sb_free(&b); sb_free(&cl);
    kmm_free(&exon); kmm_free(&intron); kmm_free(&donT); kmm_free(&donF); kmm_free(&accT); kmm_free(&accF);
    return 0;
}

/* ------------------------------------------------------------------ */
/* writers                                                            */
/* ------------------------------------------------------------------ */
static char strand_chr(int s) { return s == ORC_STRAND_POS ? '+' : s == ORC_STRAND_NEG ? '-' : '?'; }
static char css_chr(int c) { return c == ORC_CSS_CANONICAL ? 'C' : c == ORC_CSS_SEMI ? 'S' : 'N'; }

static const char *TAB_HEADER =
    "index\trefid\trefname\treflen\tstart\tend\tsize\tleft\tright\tread-strand\tss-strand\tconsensus-strand\tss1\tss2\t"
    "canonical_ss\tscore\tsuspicious\tpfp\tnb_raw_aln\tnb_dist_aln\tnb_us_aln\tnb_ms_aln\tnb_um_aln\tnb_mm_aln\tnb_bpp_aln\t"
    "nb_ppp_aln\tnb_rel_aln\trel2raw\tnb_r1_pos\tnb_r1_neg\tnb_r2_pos\tnb_r2_neg\tentropy\tmean_mismatches\tmean_readlen\t"
    "max_min_anc\tmaxmmes\tintron_score\thamming5p\thamming3p\tcoding\tpws\tsplice_sig\tuniq_junc\tprimary_junc\tnb_up_juncs\t"
    "nb_down_juncs\tdist_2_up_junc\tdist_2_down_junc\tdist_nearest_junc\tmm_score\tcoverage\tup_aln\tdown_aln\tnb_samples\t"
    "JAD01\tJAD02\tJAD03\tJAD04\tJAD05\tJAD06\tJAD07\tJAD08\tJAD09\tJAD10\tJAD11\tJAD12\tJAD13\tJAD14\tJAD15\tJAD16\tJAD17\t"
    "JAD18\tJAD19\tJAD20";

/* operator<<(ostream&, Junction&), junction.hpp:1260-1319: default ostream => %g, bools 0/1 */
char *orc_write_tab(const orc_row *rows, int64_t n, const char *const *ref_names,
                    const int32_t *ref_lens, size_t *len_out) {
    sbuf s = {0};
    sb_printf(&s, "%s\n", TAB_HEADER);
    for (int64_t i = 0; i < n; i++) {
        const orc_row *r = &rows[i];
        sb_printf(&s, "%u\t%d\t%s\t%d\t%d\t%d\t%u\t%d\t%d\t%c\t%c\t%c\t", r->id, r->refid, ref_names[r->refid],
                  ref_lens[r->refid], r->start, r->end, (uint32_t)(r->end - r->start + 1), r->left, r->right,
                  strand_chr(r->read_strand), strand_chr(r->ss_strand), strand_chr(r->cons_strand));
        sb_put(&s, (const char *)r->da1, 2);
        sb_put(&s, "\t", 1);
        sb_put(&s, (const char *)r->da2, 2);
        sb_printf(&s, "\t%c\t0\t%d\t%d\t%u\t%u\t%u\t%u\t%u\t%u\t%u\t%u\t%u\t%g\t%u\t%u\t%u\t%u\t%g\t%g\t%g\t%u\t%u\t0\t%u\t%u\t0\t0\t0\t%d\t%d\t%u\t%u\t%u\t%u\t%u\t%g\t%g\t%u\t%u\t1",
                  css_chr(r->canonical), r->suspicious, r->pfp, r->nb_raw, r->nb_dist, r->nb_raw - r->nb_ms,
                  r->nb_ms, r->nb_um, r->nb_raw - r->nb_um, r->nb_bpp, r->nb_ppp, r->nb_rel,
                  (double)r->nb_rel / (double)r->nb_raw, r->r1pos, r->r1neg, r->r2pos, r->r2neg, r->entropy,
                  r->mean_mismatches, r->mean_readlen, r->max_min_anc, r->maxmmes, r->hamming5p, r->hamming3p,
                  r->uniq, r->primary, r->nb_up_juncs, r->nb_down_juncs, r->dist_up, r->dist_down, r->dist_nearest,
                  r->mm_score, r->coverage, r->up_aln, r->down_aln);
        for (int k = 0; k < 20; k++) sb_printf(&s, "\t%u", r->jad[k]);
        sb_put(&s, "\n", 1);
    }
    sb_put(&s, "\n", 1); /* saveAll streams `(*this) << endl`, junction_system.cc:356 */
    *len_out = s.n;
    return s.p;
}

/* JunctionSystem::outputBED junction_system.cc:411-418 + Junction::outputBED junction.cc:1189-1214 */
char *orc_write_bed(const orc_row *rows, int64_t n, const char *const *ref_names, const char *source,
                    const char *version, size_t *len_out) {
    sbuf s = {0};
    sb_printf(&s, "track name=\"junctions\" description=\"Portcullis V%s junctions\"\n", (version && version[0]) ? version : "X.X.X");
    for (int64_t i = 0; i < n; i++) {
        const orc_row *r = &rows[i];
        char strand = r->cons_strand == ORC_STRAND_UNK ? '.' : strand_chr(r->cons_strand);
        sb_printf(&s, "%s\t%d\t%d\t%s_%u\t%.3f\t%c\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d\n", ref_names[r->refid], r->left,
                  r->right + 1, source, r->id, (double)r->nb_raw, strand, r->start, r->end + 1, r->start - r->left,
                  r->right - r->end, r->end - r->left + 1);
    }
    *len_out = s.n;
    return s.p;
}

/* Junction::outputIntronGFF, junction.cc:1102-1129 */
char *orc_write_intron_gff(const orc_row *rows, int64_t n, const char *const *ref_names,
                           const char *source, size_t *len_out) {
    sbuf s = {0};
    sb_reserve(&s, 1);
    s.p[0] = 0;
    for (int64_t i = 0; i < n; i++) {
        const orc_row *r = &rows[i];
        char strand = r->cons_strand == ORC_STRAND_UNK ? '?' : strand_chr(r->cons_strand);
        sb_printf(&s, "%s\t%s\tintron\t%d\t%d\t%u\t%c\t.\tmult=%u;grp=junc_%u;src=E\n", ref_names[r->refid], source,
                  r->start + 1, r->end + 1, r->nb_raw, strand, r->nb_raw, r->id);
    }
    *len_out = s.n;
    return s.p;
}

/* Junction::outputJunctionGFF + condensedOutputDescription, junction.cc:1082-1183: entropy is
 * printed with setprecision(4) in the Note and with the precision 9 that is left on the stream in
 * the trailing attributes; bools are boolalpha there. */
char *orc_write_exon_gff(const orc_row *rows, int64_t n, const char *const *ref_names,
                         const char *source, size_t *len_out) {
    static const char *STRAND_STR[] = {"POSITIVE", "NEGATIVE", "UNKNOWN"};
    static const char *CSS_STR[] = {"Canonical", "Semi-canonical", "No"};
    sbuf s = {0};
    sb_reserve(&s, 1);
    s.p[0] = 0;
    for (int64_t i = 0; i < n; i++) {
        const orc_row *r = &rows[i];
        char strand = r->cons_strand == ORC_STRAND_UNK ? '?' : strand_chr(r->cons_strand);
        uint32_t ham = r->hamming3p < r->hamming5p ? r->hamming3p : r->hamming5p;
        sb_printf(&s, "%s\t%s\tmatch\t%d\t%d\t0.0\t%c\t.\tID=junc_%u;Name=junc_%u;Note=cov:%u|rel:%u|ent:%.4g|maxmmes:%u|ham:%u;mult=%u;grp=junc_%u;src=E;",
                  ref_names[r->refid], source, r->left + 1, r->right + 1, strand, r->id, r->id, r->nb_raw, r->nb_rel,
                  r->entropy, r->maxmmes, ham, r->nb_raw, r->id);
        sb_printf(&s, "Strand: %s;Canonical?=%s;Score=0;NbAlignments=%u;NbDistinct=%u;NbReliable=%u;Entropy=%.9g;MaxMMES=%u;HammingDistance5=%u;HammingDistance3=%u;UniqueJunction=%s;PrimaryJunction=%s;\n",
                  STRAND_STR[r->cons_strand], CSS_STR[r->canonical], r->nb_raw, r->nb_dist, r->nb_rel, r->entropy,
                  r->maxmmes, r->hamming5p, r->hamming3p, r->uniq ? "true" : "false", r->primary ? "true" : "false");
        sb_printf(&s, "%s\t%s\tmatch_part\t%d\t%d\t0.0\t%c\t.\tID=junc_%u_left;Parent=junc_%u\n", ref_names[r->refid], source,
                  r->left + 1, r->start, strand, r->id, r->id);
        sb_printf(&s, "%s\t%s\tmatch_part\t%d\t%d\t0.0\t%c\t.\tID=junc_%u_right;Parent=junc_%u\n", ref_names[r->refid], source,
                  r->end + 2, r->right + 1, strand, r->id, r->id);
    }
    *len_out = s.n;
    return s.p;
}

void orc_free_text(char *p) { free(p); }
