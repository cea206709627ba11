/*
 * gtx_oracle.c -- CPU ORACLE for the genomic_overlaps count / genomic_scans counts path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is a from-scratch plain-C restatement of the
 * reference's CPU algorithms, written only so that the HIP path can be checked against
 * something that follows the reference step by step.  Nothing under the product package
 * may include, link, import or execute it; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do (as the checker, never as the thing measured or shipped).
 *
 * PARITY PINNING: the reference itself cannot be built in this image (its core.h:10 and
 * genomic_scans.cpp:15-18 include GSL headers that are not installed, and writing stand-in
 * headers is not allowed), and the reference ships no tests or golden outputs.  The oracle
 * is therefore pinned ONLY by the known-answer vectors recorded in SURVEY.md section 8(c)
 * (G2 boundary/strand toy, G3 label weights, G4/G5 ingest + error quirks; committed under
 * tests/golden/) plus the internal cross-check that its two independently restated
 * algorithms (bin index vs sorted merge; unsorted vs sorted scanner) agree.  Anything
 * outside those vectors is "parity unpinned" -- see DESIGN.md.
 *
 * All file:line citations are into /root/reference/gtools/.
 *
 * Build:  make -C oracle      (libgtx_oracle.so for ctypes + gtx_oracle CLI)
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <zlib.h>
#include <math.h>

/* ------------------------------------------------------------------------------------------ */
/* errors: the reference prints to stderr and exit(1)s (genomic_intervals.cpp:1001-1006).     */
/* The library entry points must not kill the Python test process, so errors unwind through   */
/* a message buffer + status code; the CLI main() turns them back into stderr + exit(1).      */
/* ------------------------------------------------------------------------------------------ */
static char g_err[1024];
/* per-query iteration instead of the per-index-region sums (GetOverlap/NextOverlap loop of a caller, genomic_intervals.cpp:5224-5248;
 * CountQueryOverlaps :5291-5296, CalcQueryCoverage :5254-5263): 0 = off, 1 = print every accepted (query line, index label) pair in
 * the order the reference's iterators deliver them, 2 = print per query the two sums over its overlaps (label values of the INDEX
 * regions; overlap length x label value) */
static int g_pairs_mode;
static unsigned long g_q_count, g_q_cover;
static int  g_failed;
#define FAIL(...) do { if (!g_failed) { snprintf(g_err, sizeof g_err, __VA_ARGS__); g_failed = 1; } } while (0)

const char *orc_last_error(void) { return g_err; }

static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(2); } return p; }
static char *xstrdup(const char *s) { size_t n = strlen(s) + 1; char *p = xmalloc(n); memcpy(p, s, n); return p; }

/* ------------------------------------------------------------------------------------------ */
/* tokenizer -- core.cpp:577-625                                                              */
/* ------------------------------------------------------------------------------------------ */

/* core.cpp:579-593: leading blanks are skipped before every token, a token ends at the
 * delimiter, and a trailing delimiter does not open an empty token. */
static int count_tokens(const char *s, char delim)
{
  if (!s) return 0;
  size_t k = 0; int n = 0;
  while (s[k] == ' ') k++;
  for (;;) {
    if (!s[k]) return n;
    while (s[k] && s[k] != delim) k++;
    if (s[k] == delim) k++;
    n++;
    while (s[k] == ' ') k++;
    if (!s[k]) return n;
  }
}

/* core.cpp:613-625: destructive tokenizer; returns the token, advances *cursor */
static char *next_token(char **cursor, char delim)
{
  char *b = *cursor;
  while (*b == ' ') b++;
  size_t k = 0;
  while (b[k] && b[k] != delim) k++;
  if (!b[k]) *cursor = b + k;
  else { b[k] = 0; *cursor = b + k + 1; }
  return b;
}

/* ------------------------------------------------------------------------------------------ */
/* line reader -- core.cpp:241-259 (text), 318-349 (gz), 1757-1775 (type sniff)               */
/* A line is delivered only if it ended in '\n': hitting EOF inside the read drops it         */
/* (core.cpp:243 / 253, and 333 / 343 for gz).                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  FILE *fp; gzFile gz; int is_stdin;
  char *buf; size_t cap;
  long n_line;           /* 0 = nothing current (core.cpp:153-157) */
} orc_reader;

static int reader_open(orc_reader *r, const char *path)
{
  memset(r, 0, sizeof *r);
  r->cap = 1 << 16; r->buf = xmalloc(r->cap);
  if (!path) { r->fp = stdin; r->is_stdin = 1; return 0; }
  FILE *f = fopen(path, "rb");
  if (!f) { FAIL("[CreateFileBuffer] Error: cannot open file '%s'!", path); return -1; }   /* core.cpp:450 */
  int b1 = fgetc(f), b2 = fgetc(f);
  fclose(f);
  if (b1 == 0x1f && b2 == 0x8b) {                      /* core.cpp:1765: gzip magic */
    r->gz = gzopen(path, "rb");
    if (!r->gz) { FAIL("[CreateFileBuffer] Error: cannot open file '%s'!", path); return -1; }
  } else {
    r->fp = fopen(path, "r");
    if (!r->fp) { FAIL("[FileBuffer] Error: cannot open file '%s' for reading!", path); return -1; }
  }
  return 0;
}

static void reader_close(orc_reader *r)
{
  if (r->gz) gzclose(r->gz);
  if (r->fp && !r->is_stdin) fclose(r->fp);
  free(r->buf);
  memset(r, 0, sizeof *r);
}

/* returns current line (without '\n') or NULL at end; mirrors FileBuffer*::Next */
static char *reader_next(orc_reader *r)
{
  size_t len = 0;
  for (;;) {
    if (r->cap - len < 2) { r->cap *= 2; r->buf = realloc(r->buf, r->cap); if (!r->buf) exit(2); }
    char *got = r->gz ? gzgets(r->gz, r->buf + len, (int)(r->cap - len))
                      : fgets(r->buf + len, (int)(r->cap - len), r->fp);
    if (!got) { r->n_line = 0; return NULL; }
    len += strlen(r->buf + len);
    if (len && r->buf[len - 1] == '\n') break;
    /* no newline yet: either the buffer was too small (loop and grow) or EOF cut the line */
    int at_eof = r->gz ? gzeof(r->gz) : feof(r->fp);
    if (at_eof) { r->n_line = 0; return NULL; }       /* last line without '\n' is dropped */
  }
  r->buf[len - 1] = 0;
  r->n_line++;
  return r->buf;
}

static char *reader_get(orc_reader *r) { return r->n_line > 0 ? r->buf : NULL; }

/* ------------------------------------------------------------------------------------------ */
/* data model: a region is an ordered list of intervals on one chromosome/strand               */
/* (genomic_intervals.h:91, 591).  Chromosomes are interned; ordering is strcmp order         */
/* (genomic_intervals.cpp:1227) -- in packed mode the id order IS that order by contract.      */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int   chrom;
  char  strand;
  int   n_iv;
  long  iv1[2];          /* the interval when n_iv == 1 (start,stop; 1-based inclusive) */
  long *ivn;             /* 2*n_iv longs when n_iv > 1 (BED12 blocks) */
  char *label;           /* NULL in packed mode: the weight is in `weight` */
  long  weight;
  long  n_line;
} orc_region;

/* interval array of a region, valid wherever the struct has been copied to */
#define RIV(r) ((r)->n_iv <= 1 ? (r)->iv1 : (r)->ivn)

typedef struct { char **names; int n, cap; int packed; } orc_chroms;

static int chrom_intern(orc_chroms *c, const char *name)
{
  for (int i = 0; i < c->n; i++) if (strcmp(c->names[i], name) == 0) return i;   /* few dozen names */
  if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 64; c->names = realloc(c->names, c->cap * sizeof(char *)); }
  c->names[c->n] = xstrdup(name);
  return c->n++;
}

static int chrom_cmp(const orc_chroms *c, int a, int b)
{
  if (a == b) return 0;
  if (c->packed) return a < b ? -1 : 1;
  return strcmp(c->names[a], c->names[b]);
}

static void region_free(orc_region *r)
{
  free(r->ivn);
  free(r->label);
  memset(r, 0, sizeof *r);
}

static long front_start(const orc_region *r) { return RIV(r)[0]; }
static long back_stop(const orc_region *r)  { return RIV(r)[2 * (r->n_iv - 1) + 1]; }

/* genomic_intervals.cpp:5956-5962 */
static int process_strand(const char *t, char *out)
{
  if (!strcmp(t, "1") || !strcmp(t, "+")) { *out = '+'; return 0; }
  if (!strcmp(t, "-1") || !strcmp(t, "-")) { *out = '-'; return 0; }
  if (!strcmp(t, ".")) { *out = '+'; return 0; }
  FAIL("Error: invalid strand '%s'!", t);
  return -1;
}

/* BED line -> region.  genomic_intervals.cpp:2157-2182.  `line` is clobbered. */
static int bed_parse(orc_chroms *chroms, char *line, long n_line, orc_region *out)
{
  memset(out, 0, sizeof *out);
  out->n_line = n_line;
  char sep = strchr(line, '\t') ? '\t' : ' ';                                   /* :2159 */
  int nt = count_tokens(line, sep);
  if (nt < 3) { FAIL("\nError: Line %ld: number of tokens should be at least 3 for BED format!", n_line); return -1; }
  char *p = line;
  out->chrom = chrom_intern(chroms, next_token(&p, sep));
  long start = atol(next_token(&p, sep)) + 1;                                   /* :2163 0-based -> 1-based */
  long stop  = atol(next_token(&p, sep));
  out->strand = '+';
  out->label = xstrdup(nt == 3 ? "_" : next_token(&p, sep));                    /* :2166 */
  if (nt >= 5) (void)next_token(&p, sep);                                       /* score */
  if (nt >= 6 && process_strand(next_token(&p, sep), &out->strand)) return -1;
  if (nt >= 8) { (void)next_token(&p, sep); (void)next_token(&p, sep); }        /* thickStart/End */
  if (nt >= 9) (void)next_token(&p, sep);                                       /* itemRgb */
  if (nt != 12) { out->n_iv = 1; out->iv1[0] = start; out->iv1[1] = stop; return 0; }
  /* BED12 blocks (:2174-2181): stop_k = size_k; start_k = start + off_k; stop_k += start_k - 1 */
  long nb = atol(next_token(&p, sep));
  if (nb <= 0) { FAIL("oracle: BED12 line %ld without blocks is outside the restated path", n_line); return -1; }
  long *iv = xmalloc(sizeof(long) * 2 * nb);
  char *sizes = next_token(&p, sep);
  for (long k = 0; k < nb; k++) { iv[2 * k] = start; iv[2 * k + 1] = atol(next_token(&sizes, ',')); }
  char *offs = next_token(&p, sep);
  for (long k = 0; k < nb; k++) { iv[2 * k] += atol(next_token(&offs, ',')); iv[2 * k + 1] += iv[2 * k] - 1; }
  out->n_iv = (int)nb;
  if (nb == 1) { out->iv1[0] = iv[0]; out->iv1[1] = iv[1]; free(iv); } else out->ivn = iv;
  return 0;
}

/* genomic_intervals.cpp:1081-1085 */
static long label_value(const orc_region *r, long max_label_value)
{
  if (max_label_value <= 1) return 1;
  long v = r->label ? atol(r->label) : r->weight;
  return v < max_label_value ? v : max_label_value;
}

/* genomic_intervals.cpp:1121-1161: intervals of one region must share chrom/strand (always
 * true for BED-made regions), be start-sorted and pairwise disjoint */
static int region_sorted_nonoverlapping(const orc_region *r)
{
  const long *iv = RIV(r);
  for (int k = 1; k < r->n_iv; k++) {
    if (iv[2 * k] < iv[2 * (k - 1)]) return 0;
    if (iv[2 * k] <= iv[2 * (k - 1) + 1]) return 0;
  }
  return 1;
}

/* genomic_intervals.cpp:624-630 + 1167-1172: any interval pair intersects */
static int region_overlaps(const orc_chroms *c, const orc_region *a, const orc_region *b, int ignore_strand)
{
  if (chrom_cmp(c, a->chrom, b->chrom)) return 0;
  if (!ignore_strand && a->strand != b->strand) return 0;
  const long *x = RIV(a), *y = RIV(b);
  for (int i = 0; i < a->n_iv; i++)
    for (int j = 0; j < b->n_iv; j++)
      if (!(x[2 * i] > y[2 * j + 1] || x[2 * i + 1] < y[2 * j])) return 1;
  return 0;
}

/* genomic_intervals.cpp:1207-1236: +1 = r lies entirely before q, -1 = q entirely before r,
 * 0 = envelopes meet; chromosome (and, if asked, strand) order first */
static int region_direction(const orc_chroms *c, const orc_region *q, int r_chrom, char r_strand, long r_start, long r_stop, int by_strand)
{
  int d = chrom_cmp(c, q->chrom, r_chrom);
  if (d) return d;
  if (by_strand) { int s = (int)q->strand - (int)r_strand; if (s) return s; }
  if (r_stop < front_start(q)) return 1;
  if (back_stop(q) < r_start) return -1;
  return 0;
}

/* genomic_intervals.cpp:396-401 on the regions' first intervals (:1177-1180) */
static int region_is_before(const orc_chroms *c, const orc_region *a, const orc_region *b, int by_strand)
{
  int d = chrom_cmp(c, a->chrom, b->chrom);
  if (d) return d < 0;
  if (by_strand && a->strand != b->strand) return a->strand < b->strand;
  return front_start(a) < front_start(b);
}

/* genomic_intervals.cpp:427-432 + 1196-1202: sum over interval pairs of max(0, min(stops)-max(starts)+1),
 * 0 for a pair on different chromosomes / (unless ignored) strands */
static long region_calc_overlap(const orc_chroms *c, const orc_region *a, const orc_region *b, int ignore_strand)
{
  if (chrom_cmp(c, a->chrom, b->chrom)) return 0;
  if (!ignore_strand && a->strand != b->strand) return 0;
  const long *x = RIV(a), *y = RIV(b);
  long tot = 0;
  for (int i = 0; i < a->n_iv; i++)
    for (int j = 0; j < b->n_iv; j++) {
      long lo = x[2 * i] > y[2 * j] ? x[2 * i] : y[2 * j], hi = x[2 * i + 1] < y[2 * j + 1] ? x[2 * i + 1] : y[2 * j + 1];
      if (hi - lo + 1 > 0) tot += hi - lo + 1;
    }
  return tot;
}

/* what one overlapping (query, index) pair adds: its label value for `count` (genomic_intervals.cpp:5312),
 * overlap length x label value for `coverage` (:5278-5280) */
static uint64_t pair_value(const orc_chroms *c, const orc_region *q, const orc_region *r, int coverage, int match_gaps, int ignore_strand, long w)
{
  if (!coverage) return (uint64_t)w;
  long cc;
  if (match_gaps) {
    long lo = front_start(q) > front_start(r) ? front_start(q) : front_start(r), hi = back_stop(q) < back_stop(r) ? back_stop(q) : back_stop(r);
    cc = hi - lo + 1;
  } else cc = region_calc_overlap(c, r, q, ignore_strand);
  cc *= w;
  return (uint64_t)cc;
}

static void per_query_pair(const orc_chroms *c, const orc_region *q, const orc_region *r, int match_gaps, int ignore_strand, long max_label_value)
{
  if (g_pairs_mode == 1) { printf("%ld\t%s\n", q->n_line, r->label ? r->label : "_"); return; }
  long lv = label_value(r, max_label_value);
  long cc;
  if (match_gaps) {
    long lo = front_start(q) > front_start(r) ? front_start(q) : front_start(r), hi = back_stop(q) < back_stop(r) ? back_stop(q) : back_stop(r);
    cc = hi - lo + 1;
  } else cc = region_calc_overlap(c, q, r, ignore_strand);
  g_q_count += (unsigned long)lv; g_q_cover += (unsigned long)(cc * lv);
}

/* genomic_intervals.cpp:5224-5248: filter applied to every candidate */
static int accept_overlap(const orc_chroms *c, const orc_region *q, const orc_region *r, int match_gaps, int ignore_strand)
{
  if (!(match_gaps || region_overlaps(c, q, r, ignore_strand))) return 0;
  return ignore_strand || q->strand == r->strand;
}

/* ------------------------------------------------------------------------------------------ */
/* query sources: a BED stream (genomic_intervals.cpp:3855-3861) or a packed triple array      */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  orc_chroms *chroms;
  orc_reader *rd; int started;                          /* text */
  const int32_t *tri; const int32_t *w; int64_t n, i;   /* packed */
} orc_source;

/* skip "browser "/"track " header lines (genomic_intervals.cpp:3713-3720); leaves the first data line current */
static void skip_header(orc_reader *rd)
{
  char *l = reader_next(rd);
  while (l && (strncmp(l, "browser ", 8) == 0 || strncmp(l, "track ", 6) == 0)) l = reader_next(rd);
}

/* 1 = produced, 0 = end, -1 = error */
static int source_next(orc_source *s, orc_region *out)
{
  if (s->rd) {
    if (!s->started) { skip_header(s->rd); s->started = 1; }
    char *l = reader_get(s->rd);
    if (!l) return 0;
    if (bed_parse(s->chroms, l, s->rd->n_line, out)) return -1;
    reader_next(s->rd);                                                         /* :2122 B->Next() */
    return 1;
  }
  if (s->i >= s->n) return 0;
  const int32_t *t = s->tri + 3 * s->i;
  memset(out, 0, sizeof *out);
  out->chrom = t[0]; out->strand = '+'; out->n_iv = 1;
  out->iv1[0] = t[1]; out->iv1[1] = t[2]; out->n_line = (long)s->i + 1;
  out->weight = s->w ? s->w[s->i] : 1;
  s->i++;
  return 1;
}

/* in-memory index set (load_in_memory = true, genomic_intervals.cpp:3784-3817) */
typedef struct { orc_region *R; long n; } orc_set;

__attribute__((unused)) static int set_load_bed(orc_chroms *chroms, const char *path, orc_set *set)
{
  orc_reader rd; orc_source src; long cap = 1024;
  memset(set, 0, sizeof *set);
  if (reader_open(&rd, path)) return -1;
  memset(&src, 0, sizeof src); src.chroms = chroms; src.rd = &rd;
  set->R = xmalloc(cap * sizeof(orc_region));
  for (;;) {
    if (set->n == cap) { cap *= 2; set->R = realloc(set->R, cap * sizeof(orc_region)); }
    int k = source_next(&src, &set->R[set->n]);
    if (k < 0) { reader_close(&rd); return -1; }
    if (!k) break;
    set->n++;
  }
  reader_close(&rd);
  return 0;
}

__attribute__((unused)) static void set_free(orc_set *s) { for (long i = 0; i < s->n; i++) region_free(&s->R[i]); free(s->R); memset(s, 0, sizeof *s); }

/* ------------------------------------------------------------------------------------------ */
/* bin index -- UnsortedGenomicRegionSetOverlaps, genomic_intervals.cpp:5593-5764              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int n_levels; int bits[16];
  int n_chrom;                 /* indexed by chrom id; NULL entry = chromosome not in index */
  long **nbins;                /* [chrom][level] */
  long ***head;                /* [chrom][level][bin] -> newest region or -1 */
  long *next;                  /* chain */
} orc_binindex;

static void binindex_free(orc_binindex *bx)
{
  for (int c = 0; c < bx->n_chrom; c++) if (bx->head && bx->head[c]) {
    for (int l = 0; l < bx->n_levels; l++) free(bx->head[c][l]);
    free(bx->head[c]); free(bx->nbins[c]);
  }
  free(bx->head); free(bx->nbins); free(bx->next);
  memset(bx, 0, sizeof *bx);
}

static int binindex_build(const orc_chroms *chroms, const orc_set *set, const char *bin_bits, orc_binindex *bx)
{
  memset(bx, 0, sizeof *bx);
  /* levels (:5619-5636): default 17,20,23,26 and the last level is forced to 60 bits */
  if (!bin_bits) { bx->n_levels = 5; bx->bits[0] = 17; bx->bits[1] = 20; bx->bits[2] = 23; bx->bits[3] = 26; }
  else {
    char *tmp = xstrdup(bin_bits), *p = tmp; int l = 0;
    bx->n_levels = count_tokens(tmp, ',') + 1;
    if (bx->n_levels > 16) bx->n_levels = 16;
    for (char *t = next_token(&p, ','); t[0] && l < 15; t = next_token(&p, ',')) bx->bits[l++] = atoi(t);
    free(tmp);
  }
  bx->bits[bx->n_levels - 1] = 60;

  int nc = chroms->packed ? 0 : chroms->n;
  if (chroms->packed) for (long k = 0; k < set->n; k++) if (set->R[k].chrom + 1 > nc) nc = set->R[k].chrom + 1;
  bx->n_chrom = nc;
  long *maxstop = xmalloc(sizeof(long) * (nc + 1));
  for (int c = 0; c < nc; c++) maxstop[c] = -1;
  bx->next = xmalloc(sizeof(long) * (set->n + 1));
  bx->nbins = xmalloc(sizeof(long *) * (nc + 1));
  bx->head  = xmalloc(sizeof(long **) * (nc + 1));
  for (int c = 0; c < nc; c++) { bx->nbins[c] = NULL; bx->head[c] = NULL; }
  for (long k = 0; k < set->n; k++) {                                           /* :5603-5616 */
    const orc_region *r = &set->R[k];
    bx->next[k] = -1;
    if (!region_sorted_nonoverlapping(r)) { FAIL("\nError: Line %ld: index regions should be compatible, sorted and non-overlapping!", r->n_line); free(maxstop); return -1; }
    long s = front_start(r), e = back_stop(r);
    if (s > e || e <= 0) continue;
    if (maxstop[r->chrom] < 0 || e > maxstop[r->chrom]) maxstop[r->chrom] = e;
  }
  for (int c = 0; c < nc; c++) {                                                /* :5638-5651 */
    if (maxstop[c] < 0) continue;
    bx->nbins[c] = xmalloc(sizeof(long) * bx->n_levels);
    bx->head[c]  = xmalloc(sizeof(long *) * bx->n_levels);
    for (int l = 0; l < bx->n_levels; l++) {
      long nb = (maxstop[c] >> bx->bits[l]) + 1;
      bx->nbins[c][l] = nb;
      bx->head[c][l] = xmalloc(sizeof(long) * nb);
      for (long b = 0; b < nb; b++) bx->head[c][l][b] = -1;
    }
  }
  for (long k = 0; k < set->n; k++) {                                           /* :5655-5674 */
    const orc_region *r = &set->R[k];
    long s = front_start(r), e = back_stop(r);
    if (s > e || e <= 0) continue;
    if (s <= 0) s = 1;
    for (int l = 0; l < bx->n_levels; l++) {
      long b0 = s >> bx->bits[l], b1 = e >> bx->bits[l];
      if (b0 != b1) continue;
      long z = bx->head[r->chrom][l][b0];
      if (z != -1) bx->next[k] = z;
      bx->head[r->chrom][l][b0] = k;
      break;
    }
  }
  free(maxstop);
  return 0;
}

/* CountIndexOverlaps over the bin index: genomic_intervals.cpp:5304-5317 driving
 * GetQuery/NextQuery (:5693-5711), GetMatch/NextMatch (:5717-5764), GetOverlap (:5224-5248) */
static int count_with_binindex(orc_chroms *chroms, const orc_set *set, const orc_binindex *bx, orc_source *src,
                               int match_gaps, int ignore_strand, long max_label_value, uint64_t *hits, int coverage)
{
  for (long k = 0; k < set->n; k++) hits[k] = 0;
  orc_region q;
  int k;
  while ((k = source_next(src, &q)) == 1) {
    int rc = 0;
    if (!region_sorted_nonoverlapping(&q)) { FAIL("\nError: Line %ld: query regions should be compatible, sorted and non-overlapping!", q.n_line); rc = -1; }
    /* unknown chromosome: no bin set, so nothing is validated either (:5719-5720, :5731) */
    int known = !rc && q.chrom >= 0 && q.chrom < bx->n_chrom && bx->head[q.chrom];
    if (known) {
      long s = front_start(&q), e = back_stop(&q);
      if (e <= 0) { FAIL("\nError: Line %ld: stop position must be positive!", q.n_line); rc = -1; }                          /* :5740 */
      else if (s > e) { FAIL("\nError: Line %ld: start position cannot be greater than stop position!", q.n_line); rc = -1; } /* :5741 */
      else {
        if (s <= 0) s = 1;
        long *nb = bx->nbins[q.chrom];
        if ((s >> bx->bits[0]) < nb[0]) {                                         /* :5745 */
          long w = label_value(&q, max_label_value);
          for (int l = 0; l < bx->n_levels; l++) {
            long b0 = s >> bx->bits[l], b1 = e >> bx->bits[l];
            if (b1 > nb[l] - 1) b1 = nb[l] - 1;
            for (long b = b0; b <= b1; b++)
              for (long z = bx->head[q.chrom][l][b]; z != -1; z = bx->next[z]) {
                const orc_region *r = &set->R[z];
                if (s <= back_stop(r) && e >= front_start(r) && accept_overlap(chroms, &q, r, match_gaps, ignore_strand)) {
                  if (g_pairs_mode) per_query_pair(chroms, &q, r, match_gaps, ignore_strand, max_label_value);
                  else hits[z] += pair_value(chroms, &q, r, coverage, match_gaps, ignore_strand, w);   /* :5312 / :5280 */
                }
              }
          }
        }
      }
    }
    if (g_pairs_mode == 2 && !rc) { printf("%ld\t%lu\t%lu\n", q.n_line, g_q_count, g_q_cover); g_q_count = g_q_cover = 0; }
    region_free(&q);
    if (rc) return -1;
  }
  return k < 0 ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------ */
/* sorted merge -- SortedGenomicRegionSetOverlaps, genomic_intervals.cpp:5807-5937             */
/* ------------------------------------------------------------------------------------------ */
static int count_with_merge(orc_chroms *chroms, const orc_set *set, orc_source *src, int by_strand,
                            int match_gaps, int ignore_strand, long max_label_value, uint64_t *hits, int coverage)
{
  for (long k = 0; k < set->n; k++) hits[k] = 0;
  long *buf = xmalloc(sizeof(long) * (set->n + 1)); long nbuf = 0;               /* IRegBuffer, as index numbers */
  int u_chrom = 0; char u_strand = '+'; long u_start = 0, u_stop = 0;             /* ireg_buffer_interval */
  long ip = 0;                                                                    /* IndexSet cursor */
  orc_region q, prev; int have_prev = 0, k, rc = 0;
  memset(&prev, 0, sizeof prev); memset(&q, 0, sizeof q);
  while ((k = source_next(src, &q)) == 1) {
    if (!region_sorted_nonoverlapping(&q)) { FAIL("\nError: Line %ld: query regions should be compatible, sorted and non-overlapping!", q.n_line); rc = -1; break; }
    if (have_prev && region_is_before(chroms, &q, &prev, by_strand)) {           /* :5894 */
      FAIL("\nError: Line %ld: query regions are not sorted (sorted-by-strand = %s)!", q.n_line, by_strand ? "true" : "false"); rc = -1; break;
    }
    /* LoadIndexBuffer :5844-5873 */
    if (nbuf && region_direction(chroms, &q, u_chrom, u_strand, u_start, u_stop, by_strand) > 0) nbuf = 0;
    while (ip < set->n) {
      const orc_region *r = &set->R[ip];
      /* n_line of index regions was overwritten with their ordinal at :5309 */
      /* (a caller that only iterates never overwrites them: then they are the file's line numbers) */
      if (!region_sorted_nonoverlapping(r)) { FAIL("\nError: Line %ld: index regions should be compatible, sorted and non-overlapping!", g_pairs_mode ? r->n_line : ip); rc = -1; break; }
      int d = region_direction(chroms, &q, r->chrom, r->strand, front_start(r), back_stop(r), by_strand);
      if (d < 0) break;
      if (d == 0) {
        if (!nbuf) { u_chrom = r->chrom; u_strand = r->strand; u_start = front_start(r); u_stop = back_stop(r); }
        else { if (front_start(r) < u_start) u_start = front_start(r); if (back_stop(r) > u_stop) u_stop = back_stop(r); }
        buf[nbuf++] = ip;
      }
      ip++;
      if (ip < set->n && region_is_before(chroms, &set->R[ip], r, by_strand)) {  /* :5868 */
        FAIL("\nError: Line %ld: index regions are not sorted (sorted-by-strand = %s)!", g_pairs_mode ? set->R[ip].n_line : ip, by_strand ? "true" : "false"); rc = -1; break;
      }
    }
    if (rc) break;
    /* GetOverlap/NextOverlap over GetMatch/NextMatch :5903-5926 */
    long w = label_value(&q, max_label_value);
    long j = 0;
    while (j < nbuf) {
      const orc_region *r = &set->R[buf[j]];
      int d = region_direction(chroms, &q, r->chrom, r->strand, front_start(r), back_stop(r), by_strand);
      if (d > 0) { memmove(buf + j, buf + j + 1, sizeof(long) * (nbuf - j - 1)); nbuf--; continue; }   /* erase, stay */
      if (d < 0) break;
      if (accept_overlap(chroms, &q, r, match_gaps, ignore_strand)) {
        if (g_pairs_mode) per_query_pair(chroms, &q, r, match_gaps, ignore_strand, max_label_value);
        else hits[buf[j]] += pair_value(chroms, &q, r, coverage, match_gaps, ignore_strand, w);
      }
      j++;
    }
    if (g_pairs_mode == 2) { printf("%ld\t%lu\t%lu\n", q.n_line, g_q_count, g_q_cover); g_q_count = g_q_cover = 0; }
    if (have_prev) region_free(&prev);
    prev = q; have_prev = 1;
    memset(&q, 0, sizeof q);
  }
  region_free(&q);
  if (have_prev) region_free(&prev);
  free(buf);
  return (rc || k < 0) ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------ */
/* scanners -- genomic_intervals.cpp:4832-5141                                                 */
/* bounds: chromosome -> length, iterated in std::map<string> (strcmp) order (:5997-6015)      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int n; int *chrom; long *len; } orc_bounds;   /* in iteration order */

static int bounds_find(const orc_bounds *b, int chrom) { for (int i = 0; i < b->n; i++) if (b->chrom[i] == chrom) return i; return -1; }

/* UnsortedGenomicRegionSetScanner ctor :5019-5080.  *out receives bounds->n * n_strands arrays
 * v[0..n_windows] with v[0] = number of windows. */
static int scan_unsorted(orc_chroms *chroms, orc_source *src, const orc_bounds *b, long step, long size, long max_label_value,
                         int ignore_strand, char prep, uint64_t ***out)
{
  (void)chroms;
  if (size % step) { FAIL("Error: window size must be a multiple of window step in 'GenomicRegionSetScanner'!"); return -1; }   /* :4845 */
  long comb = size / step;
  int ns = ignore_strand ? 1 : 2;
  uint64_t **v = xmalloc(sizeof(uint64_t *) * (b->n * ns + 1));
  for (int i = 0; i < b->n; i++) for (int z = 0; z < ns; z++) {
    unsigned long n = (unsigned long)(b->len[i] / step);                        /* :5025 */
    uint64_t *a = xmalloc(sizeof(uint64_t) * (n + 1)); a[0] = n;
    for (unsigned long k = 1; k <= n; k++) a[k] = 0;
    v[i * ns + z] = a;
  }
  orc_region r; int k;
  while ((k = source_next(src, &r)) == 1) {                                       /* :5036-5055 */
    int rc = 0;
    const long *iv = RIV(&r);
    for (int i = 0; i < r.n_iv && !rc; i++) {
      long s = iv[2 * i], e = iv[2 * i + 1];
      if (s > e || e <= 0) continue;
      int bi = bounds_find(b, r.chrom);
      if (bi < 0) continue;
      long pos;
      if (prep == '1') pos = s; else if (prep == 'c') pos = s + (e - s) / 2;
      else { FAIL("Error: [UnsortedGenomicRegionSetScanner] preprocess operator '%c' not supported!", prep); rc = -1; break; }
      long w = (pos - 1) / step + 1;
      uint64_t *a = v[bi * ns + ((ignore_strand || r.strand == '+') ? 0 : 1)];
      if (pos >= 1 && w <= (long)a[0]) a[w] += (uint64_t)label_value(&r, max_label_value);
    }
    region_free(&r);
    if (rc) { k = -1; break; }
  }
  if (k < 0) { for (int i = 0; i < b->n * ns; i++) free(v[i]); free(v); return -1; }
  for (int i = 0; i < b->n * ns; i++) {                                           /* :5058-5075 */
    uint64_t *a = v[i];
    if (a[0] < (uint64_t)comb) { a[0] = 0; continue; }
    uint64_t sum = 0;
    for (long j = 1; j <= comb - 1; j++) sum += a[j];
    a[0] = a[0] - comb + 1;
    for (uint64_t j = 1; j <= a[0]; j++) { sum += a[j + comb - 1]; uint64_t c = a[j]; a[j] = sum; sum -= c; }
  }
  *out = v;
  return 0;
}

/* SortedGenomicRegionSetScanner::Next :4928-4957, restated as a generator that reports
 * (value, bounds index, strand, start, stop) rows through a callback.  Regions are pulled with the
 * sort check of GenomicRegionSet::Next(sorted_by_strand,...) (:3873-3882).  'c' is rejected by the
 * reference here (:4944).  'p' (:4939-4942, the mappability input of `peaks`) adds the part of the
 * region that lies in the micro-window and, as written there, moves on TWICE after a region that ends
 * inside the micro-window (:4940 and again :4945 -- the region behind it is never looked at) and ONCE
 * after a region that reaches beyond it, whose start it has set to stop + 1 by then (:4941): the rest of
 * that region is dropped, and the order check of the pull (:3879) compares the next region with the
 * moved start.  One case is left undefined by the reference: when the first of the two pulls meets the
 * end of the stream the second one deletes the last region again (:3880 on a pointer :3858 no longer
 * owns); here the walk simply ends. */
typedef void (*orc_emit_fn)(void *ctx, long value, int bidx, char strand, long start, long stop);
static int g_scan_err_at_open;                 /* the error came from the constructor's first read (:4882), not from a Next() */

typedef struct { orc_chroms *chroms; orc_source *src; orc_region cur; int have; int by_strand; int err; } orc_pull;

static void pull_next(orc_pull *p)
{
  orc_region nx; int k = source_next(p->src, &nx);
  if (k < 0) p->err = 1;
  if (k == 1 && p->have && region_is_before(p->chroms, &nx, &p->cur, p->by_strand)) {
    FAIL("\nError: Line %ld: input regions are not sorted (sorted-by-strand = %s)!", nx.n_line, p->by_strand ? "true" : "false");
    p->err = 1;
  }
  if (p->have) region_free(&p->cur);
  p->have = 0;
  if (k == 1) { p->cur = nx; p->have = 1; }
}

static int scan_sorted(orc_chroms *chroms, orc_source *src, const orc_bounds *b, long step, long size, long max_label_value,
                       int ignore_strand, char prep, orc_emit_fn emit, void *ctx)
{
  if (size % step) { FAIL("Error: window size must be a multiple of window step in 'GenomicRegionSetScanner'!"); return -1; }
  long comb = size / step;
  long *ring = xmalloc(sizeof(long) * comb);
  orc_pull p; memset(&p, 0, sizeof p); p.chroms = chroms; p.src = src; p.by_strand = !ignore_strand;
  pull_next(&p);                                                                   /* r = R->Get() :4884 */
  int rc = p.err ? -1 : 0;
  g_scan_err_at_open = p.err;
  for (int bi = 0; bi < b->n && !rc; bi++) {
    for (char strand = '+'; strand != ' ' && !rc; strand = (strand == '+' && !ignore_strand) ? '-' : ' ') {
      for (long j = 0; j < comb; j++) ring[j] = 0;
      /* skip regions that sort before this (chromosome,strand) block :4899-4903, :4933 */
      while (p.have && !p.err) {
        int t = chrom_cmp(chroms, b->chrom[bi], p.cur.chrom);
        if (!(t > 0 || (t == 0 && strand > p.cur.strand))) break;
        pull_next(&p);
      }
      if (p.err) { rc = -1; break; }
      long k = 0, sum = 0;
      for (long start = 1, stop = step; stop <= b->len[bi]; start += step, stop += step, k = (k + 1) % comb) {
        sum -= ring[k]; ring[k] = 0;
        while (p.have && chrom_cmp(chroms, p.cur.chrom, b->chrom[bi]) == 0 && (ignore_strand || p.cur.strand == strand) && front_start(&p.cur) <= stop) {
          if (p.cur.n_iv != 1) { FAIL("\nError: Line %ld: single-interval regions expected for this operation!\n", p.cur.n_line); rc = -1; break; }
          if (prep == 'p') {
            long *iv = p.cur.iv1;
            if (iv[1] <= stop) { ring[k] += iv[1] - iv[0] + 1; pull_next(&p); if (p.err) { rc = -1; break; } if (!p.have) break; }
            else { ring[k] += stop - iv[0] + 1; iv[0] = stop + 1; }
          }
          else if (prep == '1') ring[k] += label_value(&p.cur, max_label_value);
          else { FAIL("Error: [SortedGenomicRegionSetScanner] preprocess operator '%c' not supported!", prep); rc = -1; break; }
          pull_next(&p);
          if (p.err) { rc = -1; break; }
        }
        if (rc) break;
        sum += ring[k];
        if (stop >= size) emit(ctx, sum, bi, strand, stop - size + 1, stop);
      }
    }
  }
  if (p.have) region_free(&p.cur);
  free(ring);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* library entry points on packed int32 triples (class_id, start, end), 1-based inclusive.    */
/* class ids stand in for chromosome names (id order == strcmp order by contract), every       */
/* region is single-interval '+', so strand-aware runs are expressed by the caller's class     */
/* mapping exactly as in the C ABI (include/gtx.h).                                            */
/* ------------------------------------------------------------------------------------------ */
static void set_from_packed(const int32_t *tri, int64_t m, orc_set *set)
{
  set->n = (long)m; set->R = xmalloc(sizeof(orc_region) * (m + 1));
  for (int64_t k = 0; k < m; k++) {
    orc_region *r = &set->R[k]; memset(r, 0, sizeof *r);
    r->chrom = tri[3 * k]; r->strand = '+'; r->n_iv = 1;
    r->iv1[0] = tri[3 * k + 1]; r->iv1[1] = tri[3 * k + 2]; r->n_line = (long)k + 1;
  }
}

/* algo: 0 = bin index (genomic_overlaps count, default), 1 = sorted merge (-S).
 * weights may be NULL (every read counts 1); otherwise w_q = min(max_label_value, weights[q])
 * unless max_label_value <= 1 (then 1), as GetLabelValue does. */
static int reduce_packed(const int32_t *refs, int64_t m, const int32_t *reads, const int32_t *weights, int64_t n,
                         int algo, long max_label_value, uint64_t *hits, int coverage);

int orc_count_packed(const int32_t *refs, int64_t m, const int32_t *reads, const int32_t *weights, int64_t n,
                     int algo, long max_label_value, uint64_t *hits)
{
  return reduce_packed(refs, m, reads, weights, n, algo, max_label_value, hits, 0);
}

/* CalcIndexCoverage (genomic_intervals.cpp:5269-5285) on packed triples: sum over overlapping reads of
 * overlap length x label value */
int orc_coverage_packed(const int32_t *refs, int64_t m, const int32_t *reads, const int32_t *weights, int64_t n,
                        int algo, long max_label_value, uint64_t *cov)
{
  return reduce_packed(refs, m, reads, weights, n, algo, max_label_value, cov, 1);
}

static int reduce_packed(const int32_t *refs, int64_t m, const int32_t *reads, const int32_t *weights, int64_t n,
                         int algo, long max_label_value, uint64_t *hits, int coverage)
{
  g_failed = 0; g_err[0] = 0;
  orc_chroms ch; memset(&ch, 0, sizeof ch); ch.packed = 1;
  orc_set set; set_from_packed(refs, m, &set);
  orc_source src; memset(&src, 0, sizeof src); src.chroms = &ch; src.tri = reads; src.w = weights; src.n = n;
  long mlv = weights ? max_label_value : 1;
  int rc;
  if (algo == 0) {
    orc_binindex bx;
    rc = binindex_build(&ch, &set, NULL, &bx);
    if (!rc) rc = count_with_binindex(&ch, &set, &bx, &src, 0, 1, mlv, hits, coverage);
    binindex_free(&bx);
  } else rc = count_with_merge(&ch, &set, &src, 0, 0, 1, mlv, hits, coverage);
  free(set.R);
  return rc;
}

typedef struct { uint64_t *out; const int64_t *off; long *nwin; } packed_emit_ctx;
static void packed_emit(void *c, long value, int bidx, char strand, long start, long stop)
{
  packed_emit_ctx *e = c; (void)strand; (void)start; (void)stop;
  e->out[e->off[bidx] + e->nwin[bidx]++] = (uint64_t)value;
}

/* number of windows the scanners report for a chromosome of this length (:5061-5064) */
int64_t orc_scan_n_windows(int64_t len, int64_t win_step, int64_t win_size)
{
  int64_t n = len / win_step, c = win_size / win_step;
  return n < c ? 0 : n - c + 1;
}

/* algo: 0 = unsorted scanner, 1 = sorted scanner.  windows_out is the concatenation, per class id
 * 0..n_classes-1, of that class's orc_scan_n_windows() window sums, class c at class_offsets[c]. */
int orc_scan_packed(const int32_t *reads, const int32_t *weights, int64_t n, const int32_t *class_len, int n_classes,
                    int32_t win_step, int32_t win_size, char prep, int algo, long max_label_value,
                    uint64_t *windows_out, const int64_t *class_offsets)
{
  g_failed = 0; g_err[0] = 0;
  orc_chroms ch; memset(&ch, 0, sizeof ch); ch.packed = 1;
  orc_source src; memset(&src, 0, sizeof src); src.chroms = &ch; src.tri = reads; src.w = weights; src.n = n;
  orc_bounds b; b.n = n_classes; b.chrom = xmalloc(sizeof(int) * (n_classes + 1)); b.len = xmalloc(sizeof(long) * (n_classes + 1));
  for (int c = 0; c < n_classes; c++) { b.chrom[c] = c; b.len[c] = class_len[c]; }
  int rc;
  long mlv = weights ? max_label_value : 1;
  if (algo == 0) {
    uint64_t **v = NULL;
    rc = scan_unsorted(&ch, &src, &b, win_step, win_size, mlv, 1, prep, &v);
    if (!rc) {
      for (int c = 0; c < n_classes; c++) { for (uint64_t k = 1; k <= v[c][0]; k++) windows_out[class_offsets[c] + k - 1] = v[c][k]; free(v[c]); }
      free(v);
    }
  } else {
    long *nwin = xmalloc(sizeof(long) * (n_classes + 1)); for (int c = 0; c < n_classes; c++) nwin[c] = 0;
    packed_emit_ctx e = { windows_out, class_offsets, nwin };
    rc = scan_sorted(&ch, &src, &b, win_step, win_size, mlv, 1, prep, packed_emit, &e);
    free(nwin);
  }
  free(b.chrom); free(b.len);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* CLI: same operations, options and output format as the reference drivers                   */
/*   gtx_oracle count|rpkm|coverage|density [-S] [-s] [-i] [-gaps] [-B bits] [--max-label-value V] [-min m] REF [READS] */
/*        genomic_overlaps.cpp:185-249 (options), :408-431 (count), :438-459 (coverage), :466-490 (density),     */
/*        :746-775 (rpkm)                                                                      */
/*   gtx_oracle counts -g GENOME [-S] [-i] [-op 1|c] [-w W] [-d D] [-min m] [--max-label-value V] [READS] */
/*        genomic_scans.cpp:108-121 (options), :399-436 (RunCounts)                            */
/* ------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------ */
/* tail probabilities used by `genomic_scans peaks` (genomic_scans.cpp:317-356): what              */
/* gsl_cdf_binomial_Q / gsl_cdf_poisson_Q / gsl_cdf_ugaussian_Q stand for.  GSL is absent; these   */
/* sum the tail away from the mean term by term (saddle-point point mass, exact term ratio) and    */
/* take the complement on the other side.  Tolerance parity with GSL only (~1e-13 relative).       */
/* ------------------------------------------------------------------------------------------ */
#define ORC_EPS 2.220446049250313e-16

/* Saddle-point form of the point masses (C. Loader, "Fast and accurate computation of binomial probabilities", */
/* 2000): the large log-gamma terms are never formed, so the relative error stays ~1e-15 for any n. */
static double orc_stirling_error(double n)                  /* lgamma(n+1) - [(n + 1/2) log n - n + log(2 pi)/2] */
{
  if (n < 16.0) return lgamma(n + 1.0) - ((n + 0.5) * log(n) - n + 0.918938533204672741780329736406);
  const double n2 = n * n;
  return (1.0 / 12.0 - (1.0 / 360.0 - (1.0 / 1260.0 - (1.0 / 1680.0 - (1.0 / 1188.0) / n2) / n2) / n2) / n2) / n;
}

static double orc_deviance(double x, double np)             /* x log(x / np) + np - x, without cancellation */
{
  if (fabs(x - np) < 0.1 * (x + np)) {
    double v = (x - np) / (x + np), s = (x - np) * v, ej = 2.0 * x * v;
    v = v * v;
    for (int j = 1; j < 1000; j++) {
      ej *= v;
      const double s1 = s + ej / (2 * j + 1);
      if (s1 == s) return s1;
      s = s1;
    }
    return s;
  }
  return x * log(x / np) + np - x;
}

static double binom_mass(long k, long n, double p)    /* 0 < p < 1 */
{
  const double q = 1.0 - p;
  if (k == 0) return exp(n * log1p(-p));
  if (k == n) return exp(n * log(p));
  const double lc = orc_stirling_error((double)n) - orc_stirling_error((double)k) - orc_stirling_error((double)(n - k)) - orc_deviance((double)k, n * p) - orc_deviance((double)(n - k), n * q);
  const double lf = 1.837877066409345483560659472811 + log((double)k) + log1p(-(double)k / n);
  return exp(lc - 0.5 * lf);
}

static double pois_mass(long k, double mu)            /* mu > 0 */
{
  if (k == 0) return exp(-mu);
  return exp(-orc_stirling_error((double)k) - orc_deviance((double)k, mu)) / sqrt(6.283185307179586476925286766559 * k);
}
double orc_binomial_Q(long k, double p, long n)
{
  if (p < 0.0 || p > 1.0 || n < 0) return NAN;
  if (k < 0) return 1.0;
  if (k >= n) return 0.0;
  if (p == 0.0) return 0.0;
  if (p == 1.0) return 1.0;
  const double odds = p / (1.0 - p);
  if ((double)k + 1.0 > n * p) {
    long i = k + 1;
    double term = binom_mass(i, n, p), sum = term;
    for (; i < n; i++) { term *= (double)(n - i) / (i + 1.0) * odds; sum += term; if (term < sum * ORC_EPS) break; }
    return sum;
  }
  long i = k;
  double term = binom_mass(i, n, p), sum = term;
  for (; i > 0; i--) { term *= (double)i / (n - i + 1.0) / odds; sum += term; if (term < sum * ORC_EPS) break; }
  return 1.0 - sum;
}
double orc_poisson_Q(long k, double mu)
{
  if (mu < 0.0) return NAN;
  if (k < 0) return 1.0;
  if (mu == 0.0) return 0.0;
  if ((double)k + 1.0 > mu) {
    long i = k + 1;
    double term = pois_mass(i, mu), sum = term;
    for (;; i++) { term *= mu / (i + 1.0); sum += term; if (term < sum * ORC_EPS) break; }
    return sum;
  }
  long i = k;
  double term = pois_mass(i, mu), sum = term;
  for (; i > 0; i--) { term *= (double)i / mu; sum += term; if (term < sum * ORC_EPS) break; }
  return 1.0 - sum;
}
double orc_gaussian_Q(double x) { return 0.5 * erfc(x / M_SQRT2); }

#ifdef ORC_MAIN
static void die(void) { fflush(stdout); fprintf(stderr, "%s\n", g_err); exit(1); }

/* reference filter of `counts -r` (genomic_scans.cpp:411-420): mode 1 = Next(GenomicRegionSetIndex*) over the
 * in-memory set's bin index (genomic_intervals.cpp:4982-4991 / :5168-5178), mode 2 = Next(GenomicRegionSet*) merging
 * with a sorted streamed set (:4960-4977 / :5144-5163) */
typedef struct {
  orc_chroms *ch; const orc_bounds *b; long min_reads;
  int mode, ign, done;
  orc_set refset; orc_binindex bx; orc_pull pull;
} cli_emit_ctx;

static int window_reported(cli_emit_ctx *e, int chrom, char strand, long start, long stop)
{
  if (e->mode == 0) return 1;
  if (e->done) return 0;
  if (e->mode == 1) {                                                             /* GetOverlap(&w, false, ignore_strand) != NULL */
    if (!(chrom >= 0 && chrom < e->bx.n_chrom && e->bx.head[chrom])) return 0;
    orc_region w; memset(&w, 0, sizeof w); w.chrom = chrom; w.strand = strand; w.n_iv = 1; w.iv1[0] = start; w.iv1[1] = stop;
    long s = start <= 0 ? 1 : start, en = stop;
    long *nb = e->bx.nbins[chrom];
    if ((s >> e->bx.bits[0]) >= nb[0]) return 0;
    for (int l = 0; l < e->bx.n_levels; l++) {
      long b0 = s >> e->bx.bits[l], b1 = en >> e->bx.bits[l];
      if (b1 > nb[l] - 1) b1 = nb[l] - 1;
      for (long bb = b0; bb <= b1; bb++)
        for (long z = e->bx.head[chrom][l][bb]; z != -1; z = e->bx.next[z]) {
          const orc_region *r = &e->refset.R[z];
          if (s <= back_stop(r) && en >= front_start(r) && accept_overlap(e->ch, &w, r, 0, e->ign)) return 1;
        }
    }
    return 0;
  }
  while (e->pull.have) {                                                          /* q->CalcDirection(&w, !ignore_strand) */
    int d = region_direction(e->ch, &e->pull.cur, chrom, strand, start, stop, !e->ign);
    if (d < 0) { pull_next(&e->pull); if (e->pull.err) die(); }
    else return d == 0;
  }
  e->done = 1;                                                                    /* the set ran out: Next(Ref) answers -1 */
  return 0;
}

static void cli_emit(void *c, long value, int bidx, char strand, long start, long stop)
{
  cli_emit_ctx *e = c;
  if (!window_reported(e, e->b->chrom[bidx], strand, start, stop)) return;
  if (value >= e->min_reads) printf("%ld\t%s %c %ld %ld\n", value, e->ch->names[e->b->chrom[bidx]], strand, start, stop);   /* genomic_scans.cpp:422-426 */
}

/* ---- peaks: PeakFinder (genomic_scans.cpp:209-380) + ComputeQValues (:162-205) ---------------------- */
typedef struct { long value; int bidx; char strand; long start, stop; } win_row;
typedef struct { win_row *w; long n, cap; } win_list;
static void collect_emit(void *c, long value, int bidx, char strand, long start, long stop)
{
  win_list *l = c;
  if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 1 << 16; l->w = realloc(l->w, sizeof(win_row) * (size_t)l->cap); }
  win_row r = { value, bidx, strand, start, stop }; l->w[l->n++] = r;
}

/* CountGenomicRegions (genomic_intervals.cpp:6206-6214): a pass of its own over the file */
static long count_regions(orc_chroms *ch, const char *file, long mlv)
{
  orc_reader rd; if (reader_open(&rd, file)) die();
  orc_source src; memset(&src, 0, sizeof src); src.chroms = ch; src.rd = &rd;
  orc_region r; int k; long n = 0;
  while ((k = source_next(&src, &r)) == 1) { n += label_value(&r, mlv); region_free(&r); }
  if (k < 0) die();
  reader_close(&rd);
  return n;
}

/* `late` (sorted scanners only): an error that one of the scanner's Next() calls meets does not end the run here -- its message
 * is kept in late[] and the windows returned before it in *out, for the caller to raise it at the call that meets it */
static void scan_file(orc_chroms *ch, const orc_bounds *b, const char *file, int sorted, char prep, long dist, long win, long mlv, int ign, win_list *out, char *late)
{
  memset(out, 0, sizeof *out);
  if (late) late[0] = 0;
  orc_reader rd; if (reader_open(&rd, file)) die();
  orc_source src; memset(&src, 0, sizeof src); src.chroms = ch; src.rd = &rd;
  if (sorted) {
    if (scan_sorted(ch, &src, b, dist, win, mlv, ign, prep, collect_emit, out)) {
      if (!late || g_scan_err_at_open) die();
      snprintf(late, sizeof g_err, "%s", g_err); g_failed = 0; g_err[0] = 0;
    }
  }
  else {
    uint64_t **v;
    if (scan_unsorted(ch, &src, b, dist, win, mlv, ign, prep, &v)) die();
    int ns = ign ? 1 : 2;
    for (int i = 0; i < b->n; i++) for (int z = 0; z < ns; z++) {
      uint64_t *arr = v[i * ns + z];
      for (uint64_t k = 1; k <= arr[0]; k++) collect_emit(out, (long)arr[k], i, z ? '-' : '+', dist * ((long)k - 1) + 1, dist * ((long)k - 1) + win);
    }
  }
  reader_close(&rd);
}

static int cmp_dbl(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }

static double compute_q_values(const double *pval, const double *pval_rnd, long n, long n_permutations, double qval_cutoff)
{
  if (n == 0) return -1.0;
  double *a = xmalloc(sizeof(double) * n), *b = xmalloc(sizeof(double) * n);
  memcpy(a, pval, sizeof(double) * n); memcpy(b, pval_rnd, sizeof(double) * n);
  qsort(a, n, sizeof(double), cmp_dbl); qsort(b, n, sizeof(double), cmp_dbl);       /* list::sort */
  unsigned long *counts = xmalloc(sizeof(unsigned long) * n);
  for (long k = 0; k < n; k++) counts[k] = 0;
  long k = 0;
  for (long i = 0, j = 0; i < n && j < n; j++) {
    while (i < n && b[j] > a[i]) { i++; k++; }
    if (k < n - 1) counts[k]++;
  }
  double *q = xmalloc(sizeof(double) * n);
  for (long c = 0; c < n; c++) {
    q[c] = (float)counts[c] / n_permutations / (c + 1);
    if (c + 1 == n) break;
    counts[c + 1] += counts[c];
  }
  float min_q = q[n - 1];
  double cutoff = -1.0;
  long p = n - 1;
  for (long c = n - 2; c >= 0; c--, p--) {
    if (min_q <= qval_cutoff) { cutoff = a[p]; break; }
    if (q[c] > min_q) q[c] = min_q; else min_q = q[c];
  }
  free(a); free(b); free(counts); free(q);
  return cutoff;
}

static double max_d(double x, double y) { return x > y ? x : y; }

/* CalcRegSize (genomic_intervals.cpp:6032-6040): the sizes of the regions' intervals, summed over the file by the plain reader */
static unsigned long reg_size(orc_chroms *ch, const char *file)
{
  orc_reader rd; if (reader_open(&rd, file)) die();
  orc_source src; memset(&src, 0, sizeof src); src.chroms = ch; src.rd = &rd;
  orc_region r; int k; unsigned long n = 0;
  while ((k = source_next(&src, &r)) == 1) { const long *iv = RIV(&r); for (int i = 0; i < r.n_iv; i++) n += (unsigned long)(iv[2 * i + 1] - iv[2 * i] + 1); region_free(&r); }
  if (k < 0) die();
  reader_close(&rd);
  return n;
}

static int run_peaks(orc_chroms *ch, const orc_bounds *b, const char *signal, const char *control, const char *uniq, int sorted, long dist, long win, long mlv,
                     int ign, long min_reads, const char *method, int norm, int cmp, double pval_cut, double qval_cut)
{
  if (!control) { fprintf(stderr, "oracle: peaks without a control draws random numbers (genomic_scans.cpp:299) and is not restated\n"); return 2; }
  const char prep = (sorted || uniq) ? '1' : 'c';                                                   /* :236-237 */
  unsigned long eff = 0;
  if (uniq) eff = reg_size(ch, uniq);                                                               /* :242 */
  else for (int i = 0; i < b->n; i++) eff += (unsigned long)b->len[i];                              /* CalcBoundSize */
  fprintf(stderr, "* Effective genome size = %lu\n", eff);
  long n_signal = count_regions(ch, signal, mlv);
  double p_signal = (double)n_signal / eff;
  /* the scanners advance in lockstep (:302-303, signal first): an input error of a sorted scanner surfaces at the Next() call that
   * reads the offending line, behind the three lines below; the other scanner's may come first */
  static char s_late[sizeof g_err], c_late[sizeof g_err], u_late[sizeof g_err];
  win_list S, C, U; scan_file(ch, b, signal, sorted, prep, dist, win, mlv, ign, &S, s_late);
  long n_control = count_regions(ch, control, mlv);
  double p_control = (double)n_control / eff;
  scan_file(ch, b, control, sorted, prep, dist, win, mlv, ign, &C, c_late);
  double p_ratio = p_signal / p_control;
  fprintf(stderr, "* Signal input file = %s (reads = %lu; background probability = %.2e)\n", signal, n_signal, p_signal);
  fprintf(stderr, "* Control input file = %s (reads = %lu; background probability = %.2e)\n", control, n_control, p_control);
  fprintf(stderr, "* Signal/Control background probability = %f\n", p_ratio);
  /* the mappability track: always the sorted scanner, operator 'p', label values not used (:266-267); its first read happens here */
  memset(&U, 0, sizeof U); u_late[0] = 0;
  if (uniq) scan_file(ch, b, uniq, 1, 'p', dist, win, 1, ign, &U, u_late);
  long cap = 1024, n = 0; double *p1 = xmalloc(sizeof(double) * cap), *p2 = xmalloc(sizeof(double) * cap); long *idx = xmalloc(sizeof(long) * cap);
  for (long t = 0; ; t++) {
    if (t == S.n) { if (s_late[0]) { snprintf(g_err, sizeof g_err, "%s", s_late); die(); } break; }
    if (t == C.n && c_late[0]) { snprintf(g_err, sizeof g_err, "%s", c_late); die(); }
    if (uniq && t == U.n && u_late[0]) { snprintf(g_err, sizeof g_err, "%s", u_late); die(); }
    long v1 = S.w[t].value, v2 = C.w[t].value, v0 = uniq ? U.w[t].value : win;                      /* :300 */
    if (v1 > v0) v1 = v0;
    if (v2 > v0) v2 = v0;
    if (norm) { if (p_ratio < 1.0) v2 = (long)floor((float)v2 * p_ratio); else v1 = (long)floor((float)v1 / p_ratio); }
    if (v1 < min_reads) continue;
    double pval1, pval2;
    if (cmp) {
      if (!strcmp(method, "binomial")) {
        float pp_control = ((float)v2 + 1.0) / (v0 + 1.0);
        pval1 = orc_binomial_Q(v1, max_d(pp_control, p_signal), v0 + 1);
        float pp_signal = ((float)v1 + 1.0) / (v0 + 1.0);
        pval2 = orc_binomial_Q(v2, max_d(pp_signal, p_control), v0 + 1);
      } else if (!strcmp(method, "poisson")) {
        long pseudo = 5;
        pval1 = orc_poisson_Q(v1 + pseudo, (double)(v2 + pseudo)); pval2 = orc_poisson_Q(v2 + pseudo, (double)(v1 + pseudo));
      } else if (!strcmp(method, "binomial2")) {
        double pp_control = (double)(v2 + 1) / n_control, pp_signal = (double)(v1 + 1) / n_signal;
        pval1 = orc_binomial_Q(v1 + 1, pp_control, n_signal); pval2 = orc_binomial_Q(v2 + 1, pp_signal, n_control);
      } else if (!strcmp(method, "cbinomial")) {
        pval1 = orc_binomial_Q(v1 + 1, 0.5, v1 + v2 + 2); pval2 = orc_binomial_Q(v2 + 1, 0.5, v1 + v2 + 2);
      } else if (!strcmp(method, "normal")) {
        double pp_control = (double)(v2 + 1) / n_control, pp_signal = (double)(v1 + 1) / n_signal;
        pval1 = orc_gaussian_Q((v1 + 1 - n_signal * pp_control) / sqrt(n_signal * pp_control));
        pval2 = orc_gaussian_Q((v2 + 1 - n_control * pp_signal) / sqrt(n_control * pp_signal));
      } else { fprintf(stderr, "Error: unknown probability distribution!\n"); exit(1); }
    } else {
      if (!strcmp(method, "binomial")) { pval1 = orc_binomial_Q(v1, p_signal, v0 + 1); pval2 = orc_binomial_Q(v2, p_control, v0 + 1); }
      else if (!strcmp(method, "poisson")) { long pseudo = 5; pval1 = orc_poisson_Q(v1 + pseudo, (double)(v2 + pseudo)); pval2 = orc_poisson_Q(v2 + pseudo, (double)(v1 + pseudo)); }
      else { fprintf(stderr, "Error: unknown probability distribution!\n"); exit(1); }
    }
    if (pval1 <= pval_cut) {
      if (n == cap) { cap *= 2; p1 = realloc(p1, sizeof(double) * cap); p2 = realloc(p2, sizeof(double) * cap); idx = realloc(idx, sizeof(long) * cap); }
      p1[n] = pval1; p2[n] = pval2; idx[n] = t; n++;
    }
  }
  double cutoff = compute_q_values(p1, p2, n, 1, qval_cut);
  for (long k = 0; k < n; k++)
    if (p1[k] <= cutoff) { const win_row *w = &S.w[idx[k]]; printf("%.4e\t%s %c %ld %ld\n", p1[k], ch->names[b->chrom[w->bidx]], w->strand, w->start, w->stop); }
  return 0;
}

int main(int argc, char **argv)
{
  if (argc < 2) { fprintf(stderr, "usage: gtx_oracle count|rpkm|counts [OPTIONS] FILES\n"); return 1; }
  const char *op = argv[1]; if (op[0] == '-') op++;                                /* genomic_overlaps.cpp:180-182 */
  int sorted = 0, by_strand = 0, ign = 0, gaps = 0; long mlv = 1; const char *bits = "17,20,23,26";
  unsigned long min_count = 0; long min_reads = 10, win = 500, dist = 25; char prep = '1'; const char *genome = "", *ref_file = ""; int ref_sorted = 0;
  int is_peaks = !strcmp(op, "peaks");
  int is_scan = !strcmp(op, "counts") || is_peaks;
  const char *method = "binomial"; int norm = 0, cmp = 0; double pval_cut = 1.0, qval_cut = 0.05;
  int is_cov = !strcmp(op, "coverage") || !strcmp(op, "density");
  double min_density = 0.0;
  if (!strcmp(op, "pairs")) g_pairs_mode = 1;                                         /* checker-only operations: the iterator API */
  if (!strcmp(op, "qstats")) g_pairs_mode = 2;
  if (strcmp(op, "count") && strcmp(op, "rpkm") && !is_cov && !is_scan && !g_pairs_mode) { fprintf(stderr, "Unknown operation '%s'!\n", op); return 1; }
  int a = 2;
  for (; a < argc && argv[a][0] == '-'; a++) {                                     /* core.cpp:2420-2436 */
    const char *o = argv[a];
    #define NEEDVAL() do { if (a + 1 >= argc) { fprintf(stderr, "Error: could not set option '%s'!\n", o); return 1; } } while (0)
    if (!strcmp(o, "-v")) ;
    else if (!strcmp(o, "-S")) sorted = 1;
    else if (!strcmp(o, "-i")) ign = 1;
    else if (!is_scan && !strcmp(o, "-s")) by_strand = 1;
    else if (!is_scan && !strcmp(o, "-gaps")) gaps = 1;
    else if (!is_scan && !strcmp(o, "-B")) { NEEDVAL(); bits = argv[++a]; }
    else if (!strcmp(o, "--max-label-value")) { NEEDVAL(); mlv = atol(argv[++a]); }
    else if (!strcmp(o, "-min")) { NEEDVAL(); ++a; if (is_scan) min_reads = atol(argv[a]); else if (!strcmp(op, "count") || !strcmp(op, "coverage")) min_count = strtoul(argv[a], NULL, 10); else if (!strcmp(op, "density")) min_density = atof(argv[a]); }
    else if (is_scan && !strcmp(o, "-g")) { NEEDVAL(); genome = argv[++a]; }
    else if (is_scan && !strcmp(o, "-r")) { NEEDVAL(); ref_file = argv[++a]; }
    else if (is_scan && !strcmp(o, "-Sref")) ref_sorted = 1;
    else if (is_scan && !strcmp(o, "-w")) { NEEDVAL(); win = atol(argv[++a]); }
    else if (is_scan && !strcmp(o, "-d")) { NEEDVAL(); dist = atol(argv[++a]); }
    else if (is_scan && !is_peaks && !strcmp(o, "-op")) { NEEDVAL(); prep = argv[++a][0]; }
    else if (is_peaks && !strcmp(o, "-M")) { NEEDVAL(); method = argv[++a]; }
    else if (is_peaks && !strcmp(o, "-norm")) norm = 1;
    else if (is_peaks && !strcmp(o, "-cmp")) cmp = 1;
    else if (is_peaks && !strcmp(o, "-pval")) { NEEDVAL(); pval_cut = atof(argv[++a]); }
    else if (is_peaks && !strcmp(o, "-qval")) { NEEDVAL(); qval_cut = atof(argv[++a]); }
    else if (is_peaks && !strcmp(o, "-D")) ;
    else { fprintf(stderr, "Error: unknown option '%s'!\n", o); return 1; }
  }
  orc_chroms ch; memset(&ch, 0, sizeof ch);
  if (!is_scan) {
    if (argc - a < 1) { fprintf(stderr, "usage: gtx_oracle %s [OPTIONS] REFERENCE-REGION-FILE <TEST-REGION-FILE>\n", op); return 1; }
    if (sorted && by_strand && ign) {                                              /* genomic_overlaps.cpp:305 */
      fprintf(stderr, "[Error]: the input is sorted by chromosome/strand/start (i.e. -S and -s are set), therefore the overlap algorithm can only report strand-specific results (i.e. -i cannot be set)!\n");
      return 1;
    }
    orc_set ref; if (set_load_bed(&ch, argv[a], &ref)) die();
    orc_reader rd; if (reader_open(&rd, a + 1 < argc ? argv[a + 1] : NULL)) die();
    orc_source src; memset(&src, 0, sizeof src); src.chroms = &ch; src.rd = &rd;
    uint64_t *hits = xmalloc(sizeof(uint64_t) * (ref.n + 1));
    int rc;
    if (sorted) rc = count_with_merge(&ch, &ref, &src, by_strand, gaps, ign, mlv, hits, is_cov);
    else { orc_binindex bx; rc = binindex_build(&ch, &ref, bits, &bx); if (!rc) rc = count_with_binindex(&ch, &ref, &bx, &src, gaps, ign, mlv, hits, is_cov); }
    if (rc) { fflush(stdout); die(); }
    if (g_pairs_mode) return 0;
    if (!strcmp(op, "density")) {                                                  /* genomic_overlaps.cpp:477-486 */
      for (long k = 0; k < ref.n; k++) {
        const orc_region *r = &ref.R[k];
        long size;
        if (!gaps) { size = 0; const long *iv = RIV(r); for (int i = 0; i < r->n_iv; i++) size += iv[2 * i + 1] - iv[2 * i] + 1; }
        else size = back_stop(r) - front_start(r) + 1;
        volatile double den = (double)hits[k] / size;
        if (den >= min_density) printf("%s\t%.4e\n", r->label, den);
      }
    } else if (!strcmp(op, "count") || !strcmp(op, "coverage")) {
      for (long k = 0; k < ref.n; k++) if (hits[k] >= min_count) printf("%s\t%lu\n", ref.R[k].label, (unsigned long)hits[k]);
    } else {                                                                       /* rpkm :757-772; its -min is parsed but never applied */
      unsigned long total = 0; for (long k = 0; k < ref.n; k++) total += hits[k];
      volatile double mreads = (double)total / 1000000;
      for (long k = 0; k < ref.n; k++) {
        const orc_region *r = &ref.R[k];
        long eff;
        if (!gaps) { eff = 0; const long *iv = RIV(r); for (int i = 0; i < r->n_iv; i++) eff += iv[2 * i + 1] - iv[2 * i] + 1; }   /* GetSize(skip_gaps=true) :1049-1054 */
        else eff = back_stop(r) - front_start(r) + 1;
        volatile double zero = 0.0;
        double v = eff <= 0 ? zero / zero : (double)1000 * hits[k] / eff / mreads;
        printf("%s\t%.4e\n", r->label, v);
      }
    }
    reader_close(&rd); free(hits); set_free(&ref);
    return 0;
  }
  /* counts */
  if (is_peaks && !genome[0]) genome = "genome.reg+";                              /* genomic_scans.cpp:126 */
  if (!genome[0]) { fprintf(stderr, "Error: genome region file is necessary for this operation!\n"); return 1; }
  orc_set g; if (set_load_bed(&ch, genome, &g)) die();
  orc_bounds b; b.n = 0; b.chrom = xmalloc(sizeof(int) * (g.n + 1)); b.len = xmalloc(sizeof(long) * (g.n + 1));
  for (long k = 0; k < g.n; k++) {                                                 /* ReadBounds :5997-6015 */
    if (g.R[k].n_iv != 1) { fprintf(stderr, "label = %s\n\nError: Line %ld: genome regions should be single-interval regions!\n\n", g.R[k].label, g.R[k].n_line); return 1; }
    int bi = bounds_find(&b, g.R[k].chrom);
    if (bi < 0) { b.chrom[b.n] = g.R[k].chrom; b.len[b.n] = g.R[k].iv1[1]; b.n++; }
    else if (b.len[bi] != g.R[k].iv1[1]) { fprintf(stderr, "Error: chromosome %s has multiple lengths in genome file '%s' line %ld!\n", ch.names[g.R[k].chrom], genome, k + 1); return 1; }
  }
  for (int i = 1; i < b.n; i++)                                                    /* std::map<string> iteration order */
    for (int j = i; j > 0 && strcmp(ch.names[b.chrom[j - 1]], ch.names[b.chrom[j]]) > 0; j--) {
      int tc = b.chrom[j]; b.chrom[j] = b.chrom[j - 1]; b.chrom[j - 1] = tc;
      long tl = b.len[j]; b.len[j] = b.len[j - 1]; b.len[j - 1] = tl;
    }
  if (is_peaks) {
    if (argc - a < 1) { fprintf(stderr, "usage: gtx_oracle peaks [OPTIONS] SIGNAL-REG-FILE CONTROL-REG-FILE [GENOME-UNIQ-REG-FILE]\n"); return 1; }
    return run_peaks(&ch, &b, argv[a], a + 1 < argc ? argv[a + 1] : NULL, a + 2 < argc ? argv[a + 2] : NULL, sorted, dist, win, mlv, ign, min_reads, method, norm, cmp, pval_cut, qval_cut);
  }
  orc_reader rd; if (reader_open(&rd, a < argc ? argv[a] : NULL)) die();
  orc_source src; memset(&src, 0, sizeof src); src.chroms = &ch; src.rd = &rd;
  cli_emit_ctx e; memset(&e, 0, sizeof e); e.ch = &ch; e.b = &b; e.min_reads = min_reads; e.ign = ign;
  orc_reader rrd; orc_source rsrc;
  if (ref_file[0]) {
    if (ref_sorted) {                                                              /* streamed, order checked as it is consumed */
      if (reader_open(&rrd, ref_file)) die();
      memset(&rsrc, 0, sizeof rsrc); rsrc.chroms = &ch; rsrc.rd = &rrd;
      e.mode = 2; e.pull.chroms = &ch; e.pull.src = &rsrc; e.pull.by_strand = !ign;
      pull_next(&e.pull); if (e.pull.err) die();
    } else {
      if (set_load_bed(&ch, ref_file, &e.refset)) die();
      if (binindex_build(&ch, &e.refset, "17,20,23,26", &e.bx)) die();
      e.mode = 1;
    }
  }
  if (sorted) {
    if (scan_sorted(&ch, &src, &b, dist, win, mlv, ign, prep, cli_emit, &e)) die();
  } else {
    uint64_t **v;
    if (scan_unsorted(&ch, &src, &b, dist, win, mlv, ign, prep, &v)) die();
    int ns = ign ? 1 : 2;
    for (int i = 0; i < b.n; i++) for (int z = 0; z < ns; z++) {                   /* Next/PrintInterval :5111, :5125-5141 */
      uint64_t *arr = v[i * ns + z];
      for (uint64_t k = 1; k <= arr[0]; k++) cli_emit(&e, (long)arr[k], i, z ? '-' : '+', dist * ((long)k - 1) + 1, dist * ((long)k - 1) + win);
    }
  }
  reader_close(&rd);
  return 0;
}
#endif
