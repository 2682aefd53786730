/*
 * bam2tab.c -- CPU neighbour of the end-to-end leg (TEST / BENCH INFRASTRUCTURE ONLY, like everything in oracle/).
 *
 * A prepared directory (portcullis.sorted.alignments.bam + .bai, portcullis.genome.fa + .fai) -> .junctions.tab on
 * host cores only, shaped like the reference's `junc`:
 *   - one worker thread per target sequence, targets handed out in tid order to `threads` workers
 *     (JBThreadPool, src/junction_builder.cc:236-247,459-534);
 *   - each worker opens the BAM itself, seeks to the target's first record through the .bai and reads record after
 *     record until the target ends (BamReader::setRegion / next, lib/src/bam_reader.cc:78-146; here: zlib's raw
 *     inflate per BGZF block, deps/htslib-1.3/bgzf.c:292-316, and the bam1_t field layout of sam.c's bam_read1),
 *     keeping what BamAlignment::init keeps (lib/src/bam_alignment.cc:71-100);
 *   - the records go through orc_find_juncs (the oracle's findJuncs restatement), the per-target results through
 *     orc_finalize (merge / sort / calcJunctionStats) and orc_write_tab.
 * What it is NOT: the reference binary.  The oracle port works on decoded arrays and is several times faster per
 * thread than the reference's object-per-record loop (2.6 M vs 0.185 M reads/s/thread), so this leg
 * is a LOWER bound on the reference's wall clock on the same cores.  bench.py reports it as e2e.cpu, kind "port".
 *
 *   orc_bam2tab <prep_dir> <out.tab> <threads> <orientation SE|FR|RF|FF|UNKNOWN>
 * prints one JSON line: wall seconds, threads, records, junctions, and the seconds spent inflating+parsing vs in the
 * oracle (summed over workers).
 */
#define _GNU_SOURCE
#include "portcullis_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void die(const char *msg, const char *arg) {
    fprintf(stderr, "orc_bam2tab: %s%s%s\n", msg, arg ? ": " : "", arg ? arg : "");
    exit(2);
}
static void *xrealloc(void *p, size_t n) {
    void *q = realloc(p, n ? n : 1);
    if (!q) die("out of memory", NULL);
    return q;
}
static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint64_t rd64(const uint8_t *p) { return (uint64_t)rd32(p) | (uint64_t)rd32(p + 4) << 32; }

/* ---- BGZF stream reader: sequential blocks from a file offset ---- */
typedef struct {
    int fd;
    int64_t fpos;       /* file offset of the next block */
    uint8_t *cbuf;      /* compressed read-ahead */
    size_t cbuf_n, cbuf_at;
    uint8_t blk[65536]; /* current inflated block */
    uint32_t blk_n, blk_at;
    int eof;
    z_stream zs;
    int zs_init;
} bgzf_t;
#define CBUF_CAP (4u << 20)

static void bgzf_open_at(bgzf_t *b, const char *path, int64_t coffset) {
    memset(b, 0, sizeof *b);
    b->fd = open(path, O_RDONLY);
    if (b->fd < 0) die("cannot open", path);
    b->fpos = coffset;
    b->cbuf = (uint8_t *)xrealloc(NULL, CBUF_CAP);
}
static void bgzf_close(bgzf_t *b) {
    if (b->zs_init) inflateEnd(&b->zs);
    free(b->cbuf);
    close(b->fd);
}
/* next block into b->blk; 0 at end of file */
static int bgzf_next_block(bgzf_t *b) {
    for (;;) {
        if (b->cbuf_n - b->cbuf_at < 65536 + 26 && !b->eof) { /* top the read-ahead up */
            memmove(b->cbuf, b->cbuf + b->cbuf_at, b->cbuf_n - b->cbuf_at);
            b->cbuf_n -= b->cbuf_at;
            b->cbuf_at = 0;
            while (b->cbuf_n < CBUF_CAP) {
                ssize_t r = pread(b->fd, b->cbuf + b->cbuf_n, CBUF_CAP - b->cbuf_n, b->fpos);
                if (r < 0) {
                    if (errno == EINTR) continue;
                    die("read error", NULL);
                }
                if (r == 0) {
                    b->eof = 1;
                    break;
                }
                b->cbuf_n += (size_t)r;
                b->fpos += r;
            }
        }
        const size_t avail = b->cbuf_n - b->cbuf_at;
        if (avail == 0) return 0;
        const uint8_t *h = b->cbuf + b->cbuf_at;
        if (avail < 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) die("not a BGZF block header", NULL);
        const uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
        int64_t bsize = -1;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *f = h + 12 + x;
            const uint32_t slen = f[2] | (uint32_t)f[3] << 8;
            if (f[0] == 'B' && f[1] == 'C' && slen == 2) bsize = (int64_t)(f[4] | (uint32_t)f[5] << 8) + 1;
            x += 4 + slen;
        }
        if (bsize < 0 || (size_t)bsize > avail) die("bad or truncated BGZF block", NULL);
        const uint32_t isize = rd32(h + bsize - 4);
        if (isize > 65536) die("BGZF block declares more than 64 KB", NULL);
        if (!b->zs_init) {
            if (inflateInit2(&b->zs, -15) != Z_OK) die("inflateInit2 failed", NULL);
            b->zs_init = 1;
        } else
            inflateReset(&b->zs);
        b->zs.next_in = (Bytef *)(h + 12 + xlen);
        b->zs.avail_in = (uInt)(bsize - xlen - 20);
        b->zs.next_out = b->blk;
        b->zs.avail_out = sizeof b->blk;
        const int zr = inflate(&b->zs, Z_FINISH);
        if (zr != Z_STREAM_END || b->zs.total_out != isize) die("inflate failed", NULL);
        b->cbuf_at += (size_t)bsize;
        b->blk_n = isize;
        b->blk_at = 0;
        if (isize) return 1; /* (empty blocks: the EOF marker) */
    }
}
/* n bytes of the inflated stream into dst; returns 0 at a clean end of file before the first byte */
static int bgzf_read(bgzf_t *b, uint8_t *dst, size_t n) {
    size_t got = 0;
    while (got < n) {
        if (b->blk_at == b->blk_n && !bgzf_next_block(b)) {
            if (got == 0) return 0;
            die("file ends inside a record", NULL);
        }
        size_t take = b->blk_n - b->blk_at;
        if (take > n - got) take = n - got;
        memcpy(dst + got, b->blk + b->blk_at, take);
        b->blk_at += (uint32_t)take;
        got += take;
    }
    return 1;
}

/* ---- the prepared directory ---- */
typedef struct {
    char *name;
    int32_t len;
    int64_t fa_off;
    int32_t line_b, line_w;
    uint64_t first_voff; /* ~0: no records */
} ref_t;

static char g_bam[4096], g_bai[4096], g_fa[4096], g_fai[4096];
static ref_t *g_refs;
static int32_t g_nref;
static int g_orientation;

typedef struct {
    orc_row *rows;
    int64_t n_rows;
    orc_region reg;
    int64_t n_records;
    double t_decode, t_oracle;
} result_t;
static result_t *g_res;
static int g_next_task;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

static uint8_t xs_code(const uint8_t *p, const uint8_t *end) { /* XS:A -> 0 absent/'?'/'.', 1 '+', 2 '-', 3 other (bam_aux_get's walk) */
    while (p + 3 <= end) {
        const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
        p += 3;
        const int is_xs = t0 == 'X' && t1 == 'S';
        size_t sz = 0;
        switch (ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'd': sz = 8; break;
        case 'Z': case 'H': {
            const uint8_t *q = p;
            while (q < end && *q) q++;
            sz = (size_t)(q - p) + 1;
            break;
        }
        case 'B': {
            if (p + 5 > end) return is_xs ? 3 : 0;
            const uint8_t sub = p[0];
            const uint32_t cnt = rd32(p + 1);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            sz = 5 + es * cnt;
            break;
        }
        default: return is_xs ? 3 : 0;
        }
        if (is_xs) {
            if (ty != 'A' || p >= end) return 3;
            return p[0] == '+' ? 1 : p[0] == '-' ? 2 : (p[0] == '?' || p[0] == '.') ? 0 : 3;
        }
        if (sz > (size_t)(end - p)) return 0;
        p += sz;
    }
    return 0;
}

typedef struct {
    size_t n, cap, ops, ops_cap, sb, sb_cap;
    int32_t *pos, *l_qseq, *mtid, *mpos;
    uint16_t *flag;
    uint8_t *mapq, *xs;
    uint32_t *cig_off, *cigar;
    uint64_t *seq_off;
    uint8_t *seq4;
} soa_t;

static void soa_reserve(soa_t *s, size_t more_ops, size_t more_seq) {
    if (s->n + 2 > s->cap) {
        s->cap = s->cap ? s->cap * 2 : 1 << 16;
        s->pos = (int32_t *)xrealloc(s->pos, s->cap * 4);
        s->l_qseq = (int32_t *)xrealloc(s->l_qseq, s->cap * 4);
        s->mtid = (int32_t *)xrealloc(s->mtid, s->cap * 4);
        s->mpos = (int32_t *)xrealloc(s->mpos, s->cap * 4);
        s->flag = (uint16_t *)xrealloc(s->flag, s->cap * 2);
        s->mapq = (uint8_t *)xrealloc(s->mapq, s->cap);
        s->xs = (uint8_t *)xrealloc(s->xs, s->cap);
        s->cig_off = (uint32_t *)xrealloc(s->cig_off, (s->cap + 1) * 4);
        s->seq_off = (uint64_t *)xrealloc(s->seq_off, (s->cap + 1) * 8);
    }
    if (s->ops + more_ops > s->ops_cap) {
        s->ops_cap = (s->ops + more_ops) * 2 + 1024;
        s->cigar = (uint32_t *)xrealloc(s->cigar, s->ops_cap * 4);
    }
    if (s->sb + more_seq > s->sb_cap) {
        s->sb_cap = (s->sb + more_seq) * 2 + 4096;
        s->seq4 = (uint8_t *)xrealloc(s->seq4, s->sb_cap);
    }
}
static void soa_free(soa_t *s) {
    free(s->pos); free(s->l_qseq); free(s->mtid); free(s->mpos); free(s->flag); free(s->mapq); free(s->xs);
    free(s->cig_off); free(s->cigar); free(s->seq_off); free(s->seq4);
    memset(s, 0, sizeof *s);
}

static char *load_genome(const ref_t *r) { /* faidx_fetch_seq over the whole record: the graphic characters */
    const int64_t lines = r->line_b > 0 ? (r->len + r->line_b - 1) / r->line_b : 0;
    const int64_t raw = r->len + lines * (r->line_w - r->line_b);
    char *g = (char *)xrealloc(NULL, (size_t)r->len + 1);
    const int fd = open(g_fa, O_RDONLY);
    if (fd < 0) die("cannot open", g_fa);
    struct stat sb;
    if (fstat(fd, &sb)) die("cannot stat", g_fa);
    const uint8_t *map = (const uint8_t *)mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) die("cannot map", g_fa);
    close(fd);
    const uint8_t *buf = map + r->fa_off;
    int64_t got = raw;
    if (r->fa_off + got > (int64_t)sb.st_size) got = (int64_t)sb.st_size - r->fa_off;
    int64_t n = 0;
    if (r->line_b > 0 && r->line_w >= r->line_b) { /* regular layout: whole lines at a time */
        for (int64_t at = 0; at < got && n < r->len; at += r->line_w) {
            int64_t k = r->line_b;
            if (k > r->len - n) k = r->len - n;
            if (k > got - at) k = got - at;
            memcpy(g + n, buf + at, (size_t)k);
            n += k;
        }
        for (int64_t i = 0; i < n; i++)
            if (!(g[i] > 32 && g[i] < 127)) { /* not laid out as the .fai says: filter character by character */
                n = 0;
                break;
            }
    }
    if (n != r->len) {
        n = 0;
        for (int64_t i = 0; i < got && n < r->len; i++)
            if (buf[i] > 32 && buf[i] < 127) g[n++] = (char)buf[i];
    }
    if (n != r->len) die("FASTA record shorter than its .fai length", r->name);
    g[n] = 0;
    munmap((void *)map, (size_t)sb.st_size);
    return g;
}

static void run_target(int32_t tid) {
    result_t *R = &g_res[tid];
    memset(R, 0, sizeof *R);
    R->reg.min_len = INT32_MAX;
    const ref_t *ref = &g_refs[tid];
    if (ref->first_voff == ~0ull) return;
    double t0 = now_s();
    char *genome = load_genome(ref);
    bgzf_t bz;
    bgzf_open_at(&bz, g_bam, (int64_t)(ref->first_voff >> 16));
    if (!bgzf_next_block(&bz)) die("index points past the end of the file", NULL);
    bz.blk_at = (uint32_t)(ref->first_voff & 0xffff);
    soa_t S;
    memset(&S, 0, sizeof S);
    uint8_t *rec = NULL;
    size_t rec_cap = 0;
    for (;;) {
        uint8_t b4[4];
        if (!bgzf_read(&bz, b4, 4)) break;
        const uint32_t bs = rd32(b4);
        if (bs < 32) die("bad BAM record", NULL);
        if (bs > rec_cap) {
            rec_cap = bs * 2;
            rec = (uint8_t *)xrealloc(rec, rec_cap);
        }
        if (!bgzf_read(&bz, rec, bs)) die("file ends inside a record", NULL);
        const int32_t rt = (int32_t)rd32(rec), rp = (int32_t)rd32(rec + 4);
        if (rt != tid || rp >= ref->len) break; /* hts_itr_next: the region [0, len) of this target is over */
        const uint32_t l_name = rec[8], n_cig = rec[12] | (uint32_t)rec[13] << 8;
        const int32_t l_seq = (int32_t)rd32(rec + 16);
        const size_t seq_bytes = (size_t)((l_seq + 1) / 2);
        const uint8_t *cg = rec + 32 + l_name, *sq = cg + 4 * (size_t)n_cig;
        if (32 + (size_t)l_name + 4 * (size_t)n_cig + seq_bytes + (size_t)l_seq > bs) die("bad BAM record layout", NULL);
        soa_reserve(&S, n_cig, seq_bytes);
        const size_t i = S.n;
        S.pos[i] = rp;
        S.mapq[i] = rec[9];
        S.flag[i] = (uint16_t)(rec[14] | (uint32_t)rec[15] << 8);
        S.l_qseq[i] = l_seq;
        S.mtid[i] = (int32_t)rd32(rec + 20);
        S.mpos[i] = (int32_t)rd32(rec + 24);
        S.xs[i] = xs_code(sq + seq_bytes + (size_t)l_seq, rec + bs);
        S.cig_off[i] = (uint32_t)S.ops;
        int spliced = 0;
        for (uint32_t k = 0; k < n_cig; k++) {
            const uint32_t op = rd32(cg + 4 * k);
            S.cigar[S.ops++] = op;
            spliced |= (op & 15u) == 3u;
        }
        S.seq_off[i] = S.sb;
        if (spliced) { /* only spliced alignments ever show their bases (Junction::processJunctionWindow) */
            memcpy(S.seq4 + S.sb, sq, seq_bytes);
            S.sb += seq_bytes;
        }
        S.n++;
    }
    soa_reserve(&S, 0, 0);
    S.cig_off[S.n] = (uint32_t)S.ops;
    S.seq_off[S.n] = S.sb;
    free(rec);
    bgzf_close(&bz);
    R->n_records = (int64_t)S.n;
    R->t_decode = now_s() - t0;
    t0 = now_s();
    orc_reads rd;
    rd.n = (int64_t)S.n;
    rd.pos = S.pos; rd.flag = S.flag; rd.mapq = S.mapq; rd.xs = S.xs; rd.l_qseq = S.l_qseq; rd.mtid = S.mtid; rd.mpos = S.mpos;
    rd.cig_off = S.cig_off; rd.cigar = S.cigar; rd.seq_off = S.seq_off; rd.seq4 = S.seq4;
    const int rc = orc_find_juncs(tid, ref->len, genome, &rd, g_orientation, &R->rows, &R->n_rows, &R->reg);
    if (rc) die("orc_find_juncs failed", orc_last_error());
    R->t_oracle = now_s() - t0;
    soa_free(&S);
    free(genome);
}

static void *worker(void *arg) {
    (void)arg;
    for (;;) {
        pthread_mutex_lock(&g_mu);
        const int t = g_next_task < g_nref ? g_next_task++ : -1;
        pthread_mutex_unlock(&g_mu);
        if (t < 0) return NULL;
        run_target(t);
    }
}

static uint8_t *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) die("cannot open", path);
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *p = (uint8_t *)xrealloc(NULL, (size_t)sz + 1);
    if (fread(p, 1, (size_t)sz, f) != (size_t)sz) die("short read", path);
    fclose(f);
    p[sz] = 0;
    *n = (size_t)sz;
    return p;
}

int main(int argc, char **argv) {
    if (argc != 5) die("usage: orc_bam2tab <prep_dir> <out.tab> <threads> <orientation>", NULL);
    const double t_all = now_s();
    snprintf(g_bam, sizeof g_bam, "%s/portcullis.sorted.alignments.bam", argv[1]);
    snprintf(g_bai, sizeof g_bai, "%s/portcullis.sorted.alignments.bam.bai", argv[1]);
    snprintf(g_fa, sizeof g_fa, "%s/portcullis.genome.fa", argv[1]);
    snprintf(g_fai, sizeof g_fai, "%s/portcullis.genome.fa.fai", argv[1]);
    int threads = atoi(argv[3]);
    const char *ors[5] = {"SE", "FR", "RF", "FF", "UNKNOWN"};
    g_orientation = -1;
    for (int k = 0; k < 5; k++)
        if (!strcmp(argv[4], ors[k])) g_orientation = k;
    if (g_orientation < 0) die("bad orientation", argv[4]);
    /* BAM header (BamReader::createRefList) */
    {
        bgzf_t bz;
        bgzf_open_at(&bz, g_bam, 0);
        uint8_t h[8];
        if (!bgzf_read(&bz, h, 8) || memcmp(h, "BAM\1", 4)) die("not a BAM file", g_bam);
        const uint32_t l_text = rd32(h + 4);
        uint8_t *text = (uint8_t *)xrealloc(NULL, l_text + 1);
        if (l_text && !bgzf_read(&bz, text, l_text)) die("truncated header", NULL);
        free(text);
        if (!bgzf_read(&bz, h, 4)) die("truncated header", NULL);
        g_nref = (int32_t)rd32(h);
        g_refs = (ref_t *)calloc((size_t)g_nref + 1, sizeof(ref_t));
        for (int32_t t = 0; t < g_nref; t++) {
            if (!bgzf_read(&bz, h, 4)) die("truncated header", NULL);
            const uint32_t ln = rd32(h);
            g_refs[t].name = (char *)xrealloc(NULL, ln + 1);
            if (!bgzf_read(&bz, (uint8_t *)g_refs[t].name, ln) || !bgzf_read(&bz, h, 4)) die("truncated header", NULL);
            g_refs[t].name[ln] = 0;
            g_refs[t].len = (int32_t)rd32(h);
            g_refs[t].first_voff = ~0ull;
        }
        bgzf_close(&bz);
    }
    /* .bai: the smallest chunk begin of each target (hts_itr_query's first offset for [0, len)) */
    {
        size_t n;
        uint8_t *p = slurp(g_bai, &n);
        if (n < 8 || memcmp(p, "BAI\1", 4)) die("not a BAI index", g_bai);
        size_t at = 8;
        const int32_t nr = (int32_t)rd32(p + 4);
        for (int32_t t = 0; t < nr && t < g_nref; t++) {
            const int32_t nbin = (int32_t)rd32(p + at);
            at += 4;
            for (int32_t b = 0; b < nbin; b++) {
                const uint32_t bin = rd32(p + at);
                const int32_t nch = (int32_t)rd32(p + at + 4);
                at += 8;
                for (int32_t k = 0; k < nch; k++, at += 16)
                    if (bin != 37450 && rd64(p + at) < g_refs[t].first_voff) g_refs[t].first_voff = rd64(p + at);
            }
            const int32_t nint = (int32_t)rd32(p + at);
            at += 4 + 8 * (size_t)nint;
        }
        free(p);
    }
    /* .fai */
    {
        size_t n;
        char *p = (char *)slurp(g_fai, &n);
        for (char *line = strtok(p, "\n"); line; line = strtok(NULL, "\n")) {
            char nm[1024];
            long long len, off;
            int lb, lw;
            if (sscanf(line, "%1023s %lld %lld %d %d", nm, &len, &off, &lb, &lw) != 5) continue;
            for (int32_t t = 0; t < g_nref; t++)
                if (!strcmp(nm, g_refs[t].name)) {
                    if (len != g_refs[t].len) die("FASTA and BAM disagree on the length of", nm);
                    g_refs[t].fa_off = off;
                    g_refs[t].line_b = lb;
                    g_refs[t].line_w = lw;
                }
        }
        free(p);
    }
    if (threads < 1) threads = 1;
    if (threads > g_nref) threads = g_nref; /* src/junction_builder.cc:109-112 */
    g_res = (result_t *)calloc((size_t)g_nref + 1, sizeof(result_t));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int k = 0; k < threads; k++) pthread_create(&th[k], NULL, worker, NULL);
    for (int k = 0; k < threads; k++) pthread_join(th[k], NULL);
    /* merge (src/junction_builder.cc:258-290) */
    int64_t J = 0, n_rec = 0;
    uint64_t spl = 0, unspl = 0, sum = 0;
    double t_dec = 0, t_orc = 0, longest = 0;
    for (int32_t t = 0; t < g_nref; t++) {
        J += g_res[t].n_rows;
        n_rec += g_res[t].n_records;
        spl += g_res[t].reg.spliced;
        unspl += g_res[t].reg.unspliced;
        sum += g_res[t].reg.sum_len;
        t_dec += g_res[t].t_decode;
        t_orc += g_res[t].t_oracle;
        if (g_res[t].t_decode + g_res[t].t_oracle > longest) longest = g_res[t].t_decode + g_res[t].t_oracle;
    }
    orc_row *all = (orc_row *)xrealloc(NULL, (size_t)(J + 1) * sizeof(orc_row));
    int64_t at = 0;
    for (int32_t t = 0; t < g_nref; t++) {
        if (g_res[t].n_rows) memcpy(all + at, g_res[t].rows, (size_t)g_res[t].n_rows * sizeof(orc_row));
        at += g_res[t].n_rows;
        orc_free_rows(g_res[t].rows);
    }
    orc_finalize(all, J, (double)sum / (double)(spl + unspl));
    const char **names = (const char **)calloc((size_t)g_nref + 1, sizeof(char *));
    int32_t *lens = (int32_t *)calloc((size_t)g_nref + 1, 4);
    for (int32_t t = 0; t < g_nref; t++) {
        names[t] = g_refs[t].name;
        lens[t] = g_refs[t].len;
    }
    size_t tab_n = 0;
    char *tab = orc_write_tab(all, J, names, lens, &tab_n);
    FILE *f = fopen(argv[2], "wb");
    if (!f || fwrite(tab, 1, tab_n, f) != tab_n) die("cannot write", argv[2]);
    fclose(f);
    orc_free_text(tab);
    printf("{\"wall_s\": %.3f, \"threads\": %d, \"records\": %lld, \"junctions\": %lld, \"decode_cpu_s\": %.2f, \"oracle_cpu_s\": %.2f, "
           "\"longest_target_s\": %.2f}\n",
           now_s() - t_all, threads, (long long)n_rec, (long long)J, t_dec, t_orc, longest);
    return 0;
}
