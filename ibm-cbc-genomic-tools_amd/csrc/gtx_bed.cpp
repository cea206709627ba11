// gtx_bed.cpp -- see gtx_bed.h
#include "gtx_bed.h"

#include <emmintrin.h>
#include <fcntl.h>
#include <limits.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <future>
#include <thread>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>

namespace gtxhost {
void *(*BatchArena::take)(size_t) = nullptr;
bool (*BatchArena::give)(void *) = nullptr;
static double NowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const bool kTrace = getenv("GTX_PACK_TRACE") != nullptr;


// CPUs this process may actually use: the hardware's, capped by the container's CFS quota (cgroup v2 cpu.max, v1 cfs_quota_us).
// Running more busy threads than the quota gets the whole process throttled for the rest of every 100 ms period -- measured on
// the MI355X boxes of the pool: 256 hardware threads, a quota of 16, 64 parser threads stalled for ~60 ms of every 100.
static int EffectiveCpus()
{
  static const int n = [] {
    unsigned hc = std::thread::hardware_concurrency();
    double cpus = hc ? (double)hc : 4.0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[64]; double period = 0;
      if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) cpus = std::min(cpus, atof(q) / period);
      fclose(f);
    } else {
      double quota = -1, period = 0;
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &quota) != 1) quota = -1; fclose(g); }
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &period) != 1) period = 0; fclose(g); }
      if (quota > 0 && period > 0) cpus = std::min(cpus, quota / period);
    }
    return (int)std::max(1.0, cpus);
  }();
  return n;
}

// ---------------------------------------------------------------------------------------------------
// worker pool: the packer runs two short parallel phases per 64 MB block; threads made once, not 128 per block
// ---------------------------------------------------------------------------------------------------
namespace {
class Pool {
 public:
  // fn(t) for t in [0, n): t = 0 on the caller, the rest on pool threads; returns when all are done
  static void Run(int n, const std::function<void(int)> &fn)
  {
    if (n <= 1) { if (n == 1) fn(0); return; }
    Get().Go(n, fn);
  }
 private:
  static Pool &Get() { static Pool pool; return pool; }
  std::vector<std::thread> workers_;
  std::mutex m_; std::condition_variable wake_, done_;
  const std::function<void(int)> *fn_ = nullptr; int n_ = 0, next_ = 0, left_ = 0; unsigned long gen_ = 0; bool stop_ = false;
  void Grow(int want)
  {
    while ((int)workers_.size() < want) workers_.emplace_back([this] {
      unsigned long seen = 0;
      std::unique_lock<std::mutex> lk(m_);
      for (;;) {
        wake_.wait(lk, [&] { return stop_ || (gen_ != seen && next_ < n_); });
        if (stop_) return;
        while (next_ < n_) {
          const int t = next_++;
          const std::function<void(int)> *f = fn_;
          lk.unlock(); (*f)(t); lk.lock();
          if (--left_ == 0) done_.notify_all();
        }
        seen = gen_;
      }
    });
  }
  void Go(int n, const std::function<void(int)> &fn)
  {
    std::unique_lock<std::mutex> lk(m_);
    Grow(n - 1);
    fn_ = &fn; n_ = n; next_ = 1; left_ = n - 1; gen_++;
    wake_.notify_all();
    lk.unlock();
    fn(0);
    lk.lock();
    done_.wait(lk, [&] { return left_ == 0; });
    fn_ = nullptr; n_ = 0; next_ = 0;
  }
  ~Pool() { { std::lock_guard<std::mutex> lk(m_); stop_ = true; } wake_.notify_all(); for (auto &w : workers_) w.join(); }
};
}  // namespace

// ---------------------------------------------------------------------------------------------------
// LineSource
// ---------------------------------------------------------------------------------------------------
LineSource *LineSource::Open(const char *path, std::string *err)
{
  LineSource *s = new LineSource();
  s->buf_.resize(8u << 20);
  if (!path) {
    s->fp_ = stdin; s->is_stdin_ = true; s->raw_fd_ = fileno(stdin);
#ifdef F_SETPIPE_SZ
    // a pipe: the largest buffer an unprivileged process may ask for (64 KB by default: a writer's every 64 KB wakes the reader).  What a
    // pipe delivers to a reader that touches the bytes is ~2-4 GB/s whatever the reader does (`cat f | wc -l`): moving the pages into
    // several pipes of our own (splice) and copying them out on four threads measured the same 1.9 GB/s as plain reads -- not in the build
    (void)fcntl(s->raw_fd_, F_SETPIPE_SZ, 1 << 20);
#endif
    return s;
  }
  FILE *f = fopen(path, "rb");
  if (!f) { *err = std::string("[CreateFileBuffer] Error: cannot open file '") + path + "'!"; delete s; return nullptr; }
  int b1 = fgetc(f), b2 = fgetc(f);
  fclose(f);
  if (b1 == 0x1f && b2 == 0x8b) {                        // gzip magic
    s->gz_ = gzopen(path, "rb");
    if (s->gz_) gzbuffer(s->gz_, 1u << 20);
  } else {
    s->fp_ = fopen(path, "rb");
    // a regular file can also be read in bulk with parallel pread()s (NextBlockView)
    struct stat sb;
    if (s->fp_ && fstat(fileno(s->fp_), &sb) == 0 && S_ISREG(sb.st_mode)) { s->fd_ = fileno(s->fp_); s->file_len_ = (size_t)sb.st_size; }
  }
  if (!s->gz_ && !s->fp_) { *err = std::string("[CreateFileBuffer] Error: cannot open file '") + path + "'!"; delete s; return nullptr; }
  return s;
}

LineSource *LineSource::FromFile(FILE *fp)
{
  LineSource *s = new LineSource();
  s->buf_.resize(1u << 20);
  s->fp_ = fp; s->is_stdin_ = true;                       // never closed here: the caller owns the stream
  return s;
}

LineSource::~LineSource()
{
  if (gz_) gzclose(gz_);
  if (fp_ && !is_stdin_) fclose(fp_);
}

size_t LineSource::Fill()
{
  if (eof_) return 0;
  if (pos_ > 0) { memmove(buf_.data(), buf_.data() + pos_, end_ - pos_); end_ -= pos_; pos_ = 0; }
  if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);
  size_t room = buf_.size() - end_, got;
  got = ReadStream(buf_.data() + end_, room);
  if (got == 0) eof_ = true;
  end_ += got;
  return got;
}

// One read of a stream: inflate (.gz), the descriptor (the process's stdin: whatever is there, at least a byte), or the caller's FILE*.
size_t LineSource::ReadStream(char *dst, size_t want)
{
  if (gz_) { const int g = gzread(gz_, dst, (unsigned)std::min<size_t>(want, 1u << 30)); return g > 0 ? (size_t)g : 0; }
  if (raw_fd_ >= 0) {
    for (;;) { const ssize_t g = read(raw_fd_, dst, want); if (g >= 0) return (size_t)g; if (errno != EINTR) return 0; }
  }
  return fread(dst, 1, want, fp_);
}

char *LineSource::Next()
{
  for (;;) {
    char *nl = pos_ < end_ ? (char *)memchr(buf_.data() + pos_, '\n', end_ - pos_) : nullptr;
    if (nl) {
      char *line = buf_.data() + pos_;
      *nl = 0;
      pos_ = (size_t)(nl - buf_.data()) + 1;
      line_no_++;
      return line;
    }
    if (Fill() == 0) return nullptr;                     // EOF: a last line without '\n' is dropped
  }
}

// Up to `cap` bytes of complete lines straight into dst (page-locked memory of the caller's): a regular text file by parallel preads, a
// stream (stdin, a pipe, a .gz file through inflate, a FILE* of the caller's) by straight reads behind what the line reader had buffered;
// what follows the last newline waits in buf_ for the next call.  Returns 0 at the end of the input and (size_t)-1 when not even one
// line fits (nothing is lost: the bytes are back in buf_ for NextBlock).
size_t LineSource::ReadTextInto(char *dst, size_t cap, long *first_line)
{
  if (fd_ < 0) {
    *first_line = line_no_ + 1;
    size_t n = end_ - pos_;
    if (n > cap) return (size_t)-1;
    if (n) memcpy(dst, buf_.data() + pos_, n);
    pos_ = end_ = 0;
    while (n < cap && !eof_) {
      const size_t got = ReadStream(dst + n, cap - n);
      if (got == 0) { eof_ = true; break; }
      n += got;
    }
    if (n == 0) return 0;
    const char *nl = (const char *)memrchr(dst, '\n', n);
    const size_t keep = nl ? (size_t)(nl - dst) + 1 : 0;
    if (keep == 0 && eof_) return 0;                       // only an unterminated tail is left: dropped, like the line reader does
    const size_t back = keep ? n - keep : n;                // the tail behind the last newline (or everything: a line longer than the buffer)
    if (back > buf_.size()) buf_.resize(back + (back >> 1));
    if (back) memcpy(buf_.data(), dst + (n - back), back);
    pos_ = 0; end_ = back;
    return keep ? keep : (size_t)-1;
  }

  *first_line = line_no_ + 1;
  if (!bulk_started_) {
    bulk_started_ = true;
    long at = ftell(fp_);
    file_pos_ = at < 0 ? file_len_ : (size_t)at - (end_ - pos_);
    pos_ = end_ = 0;
  }
  const size_t left = file_len_ - file_pos_;
  if (left == 0) return 0;
  const size_t want = std::min(left, cap);
  static const int maxReaders = getenv("GTX_READ_THREADS") && atoi(getenv("GTX_READ_THREADS")) > 0 ? atoi(getenv("GTX_READ_THREADS")) : std::max(2, std::min(8, EffectiveCpus() / 2));
  const int K = (int)std::min<size_t>((size_t)maxReaders, want / (4u << 20) + 1);
  std::vector<std::thread> th;
  auto rd = [&](int k) {
    size_t b0 = want * (size_t)k / K, b1 = want * (size_t)(k + 1) / K;
    while (b0 < b1) { ssize_t g = pread(fd_, dst + b0, b1 - b0, (off_t)(file_pos_ + b0)); if (g <= 0) break; b0 += (size_t)g; }
  };
  for (int k = 1; k < K; k++) th.emplace_back(rd, k);
  rd(0);
  for (auto &x : th) x.join();
  char *nl = (char *)memrchr(dst, '\n', want);
  if (nl) { const size_t n = (size_t)(nl - dst) + 1; file_pos_ += n; return n; }
  return want == left ? 0 : (size_t)-1;                  // only an unterminated tail is left | a line longer than the buffer
}

size_t LineSource::NextBlockView(std::vector<char> &block, char **view, size_t target, long *first_line)
{
  if (fd_ < 0) { size_t n = NextBlock(block, target, first_line); *view = block.data(); return n; }
  *first_line = line_no_ + 1;
  if (!bulk_started_) {
    // bytes the line-at-a-time reader has buffered but not handed out come first
    bulk_started_ = true;
    long at = ftell(fp_);
    file_pos_ = at < 0 ? file_len_ : (size_t)at - (end_ - pos_);
    pos_ = end_ = 0;
  }
  for (;;) {
    const size_t left = file_len_ - file_pos_;
    if (left == 0) return 0;
    const size_t want = std::min(left, target);
    block.resize(want);
    // parallel pread: the kernel-to-user copy is the cost of reading a cached file, so split it
    static const int maxReaders = getenv("GTX_READ_THREADS") && atoi(getenv("GTX_READ_THREADS")) > 0 ? atoi(getenv("GTX_READ_THREADS")) : std::max(2, std::min(8, EffectiveCpus() / 4));
    const int K = (int)std::min<size_t>((size_t)maxReaders, want / (4u << 20) + 1);
    std::vector<std::thread> th;
    auto rd = [&](int k) {
      size_t b0 = want * (size_t)k / K, b1 = want * (size_t)(k + 1) / K;
      while (b0 < b1) { ssize_t g = pread(fd_, block.data() + b0, b1 - b0, (off_t)(file_pos_ + b0)); if (g <= 0) break; b0 += (size_t)g; }
    };
    for (int k = 1; k < K; k++) th.emplace_back(rd, k);
    rd(0);
    for (auto &x : th) x.join();
    char *nl = (char *)memrchr(block.data(), '\n', want);
    if (nl) { const size_t n = (size_t)(nl - block.data()) + 1; file_pos_ += n; *view = block.data(); return n; }
    if (want == left) return 0;                          // only an unterminated tail is left
    target *= 2;                                         // a line longer than the block: take more
  }
}

size_t LineSource::NextBlock(std::vector<char> &block, size_t target, long *first_line)
{
  *first_line = line_no_ + 1;
  for (;;) {
    size_t have = end_ - pos_;
    if (have >= target || eof_) {
      // last newline inside the first `target` bytes (or inside everything we have)
      size_t span = std::min(have, target);
      char *base = buf_.data() + pos_;
      char *nl = span ? (char *)memrchr(base, '\n', span) : nullptr;
      if (!nl && have > span) nl = (char *)memchr(base + span, '\n', have - span);
      if (nl) {
        size_t n = (size_t)(nl - base) + 1;
        block.assign(base, base + n);
        pos_ += n;
        return n;
      }
      if (eof_) return 0;                                // only an unterminated tail is left
    }
    Fill();
  }
}

// ---------------------------------------------------------------------------------------------------
// tokenising and one BED line
// ---------------------------------------------------------------------------------------------------
int CountTokensLike(const char *s, char delim)
{
  if (!s) return 0;
  while (*s == ' ') s++;
  int n = 0;
  while (*s) {
    while (*s && *s != delim) s++;
    if (*s == delim) s++;
    n++;
    while (*s == ' ') s++;
  }
  return n;
}

static inline char *TakeToken(char **cur, char delim)
{
  char *b = *cur;
  while (*b == ' ') b++;
  char *e = b;
  while (*e && *e != delim) e++;
  if (*e) { *e = 0; *cur = e + 1; } else *cur = e;
  return b;
}

// atol for the plain decimal tokens BED carries (leading blanks, optional sign, digits)
static inline long FastAtol(const char *p)
{
  while (*p == ' ' || (*p >= '\t' && *p <= '\r')) p++;
  bool neg = false;
  if (*p == '-') { neg = true; p++; } else if (*p == '+') p++;
  unsigned long v = 0;
  while (*p >= '0' && *p <= '9') { v = v * 10 + (unsigned long)(*p - '0'); p++; }
  return neg ? -(long)v : (long)v;
}

// the same on a token that ends at `end` instead of a NUL
static inline long FastAtolTo(const char *p, const char *end)
{
  while (p < end && (*p == ' ' || (*p >= '\t' && *p <= '\r'))) p++;
  bool neg = false;
  if (p < end && *p == '-') { neg = true; p++; } else if (p < end && *p == '+') p++;
  unsigned long v = 0;
  while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + (unsigned long)(*p - '0'); p++; }
  return neg ? -(long)v : (long)v;
}

// The same for the common case -- nothing but 1..10 digits -- without a loop over the digits: the last (up to) eight go through
// one 8-byte load (padded with '0' in front) and three multiplications; anything else (sign, blank, letters, longer) takes
// FastAtolTo.  `safe_end`: bytes before it may be read.
static inline long FastDigitsTo(const char *p, const char *end, const char *safe_end)
{
  const long L = end - p;
  if (L >= 1 && L <= 10) {
    const char *q = L > 8 ? end - 8 : p;
    if (q + 8 <= safe_end) {
      long head = 0; bool ok = true;
      for (const char *h = p; h < q; h++) { const unsigned d = (unsigned)(*h - '0'); ok &= d <= 9; head = head * 10 + (long)d; }
      const int l8 = L > 8 ? 8 : (int)L;
      uint64_t x; memcpy(&x, q, 8);
      if (l8 < 8) { x <<= (8 - l8) * 8; x |= 0x3030303030303030ull >> (l8 * 8); }
      ok &= (((x & 0xF0F0F0F0F0F0F0F0ull) | (((x + 0x0606060606060606ull) & 0xF0F0F0F0F0F0F0F0ull) >> 4)) == 0x3333333333333333ull);
      if (ok) {
        x -= 0x3030303030303030ull;
        x = (x * 10) + (x >> 8);
        x = (((x & 0x000000FF000000FFull) * 0x000F424000000064ull) + (((x >> 16) & 0x000000FF000000FFull) * 0x0000271000000001ull)) >> 32;
        return head * 100000000L + (long)x;
      }
    }
  }
  return FastAtolTo(p, end);
}

BedStatus ParseBedLine(char *line, BedFields *o, char **bad)
{
  const char sep = strchr(line, '\t') ? '\t' : ' ';
  o->n_tokens = CountTokensLike(line, sep);
  if (o->n_tokens < 3) return BED_TOO_FEW_TOKENS;
  char *cur = line;
  o->chrom = TakeToken(&cur, sep);
  o->start = FastAtol(TakeToken(&cur, sep)) + 1;
  o->stop = FastAtol(TakeToken(&cur, sep));
  o->strand = '+';
  o->label = o->n_tokens == 3 ? nullptr : TakeToken(&cur, sep);
  if (o->n_tokens >= 5) (void)TakeToken(&cur, sep);     // score
  if (o->n_tokens >= 6) {
    char *t = TakeToken(&cur, sep);
    if (!strcmp(t, "1") || !strcmp(t, "+") || !strcmp(t, ".")) o->strand = '+';
    else if (!strcmp(t, "-1") || !strcmp(t, "-")) o->strand = '-';
    else { *bad = t; return BED_BAD_STRAND; }
  }
  o->block_sizes = o->block_starts = nullptr; o->n_blocks = 0;
  if (o->n_tokens == 12) {                                  // thickStart, thickEnd, itemRgb, blockCount, blockSizes, blockStarts
    (void)TakeToken(&cur, sep); (void)TakeToken(&cur, sep); (void)TakeToken(&cur, sep);
    o->n_blocks = FastAtol(TakeToken(&cur, sep));
    o->block_sizes = TakeToken(&cur, sep);
    o->block_starts = TakeToken(&cur, sep);
  }
  return BED_OK;
}

void BedBlocks(const BedFields &f, std::vector<long> *iv)
{
  iv->clear();
  char *sz = f.block_sizes, *st = f.block_starts;
  for (long k = 0; k < f.n_blocks; k++) {
    const long size = FastAtol(TakeToken(&sz, ',')), off = FastAtol(TakeToken(&st, ','));
    const long a = f.start + off;
    iv->push_back(a); iv->push_back(size + a - 1);
  }
}

// One pass over a line of the common shape -- TAB-separated, no blanks anywhere, 3..11 columns, a plain strand
// column -- doing what ParseBedLine does for it (the tokenizer rules only differ from a plain split when blanks
// are involved).  Returns the character after the line's '\n', or NULL when the line is not of that shape (the
// line is left untouched and goes through ParseBedLine) or the block ends without a '\n'.
static inline char *ParseTabbedLine(char *line, char *end, BedFields *o, size_t *chrom_len)
{
  // the line's TABs, its '\n' and any blank, 16 bytes at a time (a BED3..BED6 read line is 20-40 bytes)
  char *tab[6]; int nt = 0;
  char *p = line, *last_tab = nullptr, *nl = nullptr;
  const __m128i vt = _mm_set1_epi8('\t'), vn = _mm_set1_epi8('\n'), vs = _mm_set1_epi8(' ');
  while (!nl) {
    if (end - p >= 16) {
      const __m128i x = _mm_loadu_si128((const __m128i *)p);
      unsigned m = (unsigned)_mm_movemask_epi8(_mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(x, vt), _mm_cmpeq_epi8(x, vn)), _mm_cmpeq_epi8(x, vs)));
      while (m) {
        char *q = p + __builtin_ctz(m);
        m &= m - 1;
        const char ch = *q;
        if (ch == '\n') { nl = q; break; }
        if (ch == ' ') return nullptr;
        if (nt < 6) tab[nt] = q;
        nt++; last_tab = q;
      }
      p += 16;
    } else {
      for (; p < end; p++) {
        const char ch = *p;
        if (ch == '\n') { nl = p; break; }
        if (ch == ' ') return nullptr;
        if (ch == '\t') { if (nt < 6) tab[nt] = p; nt++; last_tab = p; }
      }
      if (!nl) return nullptr;                                     // the block ends without a '\n'
    }
  }
  p = nl;
  if (nt < 2) return nullptr;
  // tokens as CountTokens sees them: one per TAB, plus the piece after the last TAB unless it is empty
  const int n_tokens = nt + (p > last_tab + 1 ? 1 : 0);
  if (n_tokens < 3 || n_tokens > 11) return nullptr;
  char strand = '+';
  if (n_tokens >= 6) {
    const char *t = tab[4] + 1, *te = nt >= 6 ? tab[5] : p;
    const long len = te - t;
    if (len == 1 && (t[0] == '+' || t[0] == '.' || t[0] == '1')) strand = '+';
    else if ((len == 1 && t[0] == '-') || (len == 2 && t[0] == '-' && t[1] == '1')) strand = '-';
    else return nullptr;                                        // the general path words the error
  }
  o->n_tokens = n_tokens;
  o->chrom = line; *tab[0] = 0; *chrom_len = (size_t)(tab[0] - line);
  o->start = FastDigitsTo(tab[0] + 1, tab[1], end) + 1;
  o->stop = FastDigitsTo(tab[1] + 1, nt >= 3 ? tab[2] : p, end);
  o->strand = strand;
  o->label = nullptr;
  if (n_tokens >= 4) { o->label = tab[2] + 1; if (nt >= 4) *tab[3] = 0; else *p = 0; }
  return p + 1;
}

static inline void SetErrPublic(PackError *e, long line, const std::string &msg, bool no_prefix)
{
  if (!e->set) { e->set = true; e->line = line; e->msg = msg; e->no_prefix = no_prefix; }
}

// ---------------------------------------------------------------------------------------------------
// packed region files
// ---------------------------------------------------------------------------------------------------
static const char kGtxMagic[8] = {'G', 'T', 'X', 'P', 1, 0, 0, 0};
static inline size_t Pad8(size_t x) { return (x + 7) & ~(size_t)7; }

bool GtxView::IsGtx(const char *path)
{
  if (!path) return false;
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  char m[8]; const bool ok = fread(m, 1, 8, f) == 8 && memcmp(m, kGtxMagic, 8) == 0;
  fclose(f);
  return ok;
}

GtxView *GtxView::Open(const char *path, std::string *err)
{
  int fd = open(path, O_RDONLY);
  if (fd < 0) { *err = std::string("Error: cannot open file '") + path + "'!"; return nullptr; }
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < 24) { close(fd); *err = std::string("Error: '") + path + "' is not a packed region file!"; return nullptr; }
  void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { *err = std::string("Error: cannot map file '") + path + "'!"; return nullptr; }
  const char *b = (const char *)m, *e = b + st.st_size;
  GtxView *g = new GtxView; g->map_ = m; g->map_len_ = (size_t)st.st_size;
  auto bad = [&]() { *err = std::string("Error: '") + path + "' is not a valid packed region file!"; delete g; return (GtxView *)nullptr; };
  if (memcmp(b, kGtxMagic, 8) != 0) return bad();
  uint32_t n_chrom; memcpy(&n_chrom, b + 8, 4); memcpy(&g->flags, b + 12, 4); memcpy(&g->n, b + 16, 8);
  const char *p = b + 24;
  for (uint32_t c = 0; c < n_chrom; c++) {
    if (p + 2 > e) return bad();
    uint16_t len; memcpy(&len, p, 2); p += 2;
    if (p + len > e) return bad();
    g->chrom.emplace_back(p, len); p += len;
  }
  size_t off = Pad8((size_t)(p - b));
  auto take = [&](size_t bytes) -> const char * { const char *q = b + off; off = Pad8(off + bytes); return off <= (size_t)(e - b) + 7 && q + bytes <= e ? q : nullptr; };
  g->chrom_idx = (const uint16_t *)take(2 * g->n);
  g->start = (const int32_t *)take(4 * g->n);
  g->stop = (const int32_t *)take(4 * g->n);
  g->minus = (const uint8_t *)take((g->n + 7) / 8);
  g->label = (g->flags & 1) ? (const int32_t *)take(4 * g->n) : nullptr;
  if (!g->chrom_idx || !g->start || !g->stop || !g->minus || ((g->flags & 1) && !g->label)) return bad();
  for (uint64_t i = 0; i < g->n; i++) if (g->chrom_idx[i] >= n_chrom) return bad();
  return g;
}

GtxView::~GtxView() { if (map_) munmap(map_, map_len_); }

bool WriteGtx(LineSource *src, const char *out_path, PackError *err)
{
  std::vector<std::string> names; std::vector<uint16_t> cidx; std::vector<int32_t> st, en, lab; std::vector<uint8_t> minus;
  std::string last; int last_id = -1;
  bool any_label = false, top = true;
  uint64_t n = 0;
  for (char *line = src->Next(); line; line = src->Next()) {
    if (top && (strncmp(line, "browser ", 8) == 0 || strncmp(line, "track ", 6) == 0)) continue;   // genomic_intervals.cpp:3713-3720
    top = false;
    BedFields f; char *bad = nullptr;
    const BedStatus s = ParseBedLine(line, &f, &bad);
    if (s == BED_TOO_FEW_TOKENS) { SetErrPublic(err, src->line_no(), "number of tokens should be at least 3 for BED format!", false); return false; }
    if (s == BED_BAD_STRAND) { SetErrPublic(err, src->line_no(), std::string("Error: invalid strand '") + bad + "'!", true); return false; }
    if (f.n_tokens == 12) { SetErrPublic(err, src->line_no(), "multi-interval (BED12) regions are outside the MI355X counting path!", false); return false; }
    const long v = f.label ? FastAtol(f.label) : 0;
    if (f.start >= INT_MAX - 1 || f.stop >= INT_MAX - 1 || f.start <= INT_MIN + 1 || f.stop <= INT_MIN + 1 || v > INT_MAX || v < INT_MIN) {
      SetErrPublic(err, src->line_no(), "coordinate or label value does not fit the packed 32-bit representation of the MI355X path!", false); return false;
    }
    if (last_id < 0 || last != f.chrom) {
      last = f.chrom; last_id = -1;
      for (size_t c = 0; c < names.size(); c++) if (names[c] == last) { last_id = (int)c; break; }
      if (last_id < 0) {
        if (names.size() >= 65535 || last.size() > 65535) { SetErrPublic(err, src->line_no(), "too many chromosomes for a packed region file!", false); return false; }
        names.push_back(last); last_id = (int)names.size() - 1;
      }
    }
    cidx.push_back((uint16_t)last_id); st.push_back((int32_t)f.start); en.push_back((int32_t)f.stop); lab.push_back((int32_t)v);
    if ((n & 7) == 0) minus.push_back(0);
    if (f.strand == '-') minus.back() |= (uint8_t)(1u << (n & 7));
    any_label |= f.label != nullptr;
    n++;
  }
  return WriteGtxColumns(out_path, names, n, cidx.data(), st.data(), en.data(), minus.data(), any_label ? lab.data() : nullptr, err);
}

bool WriteGtxColumns(const char *out_path, const std::vector<std::string> &names, uint64_t n, const uint16_t *cidx, const int32_t *st, const int32_t *en,
                     const uint8_t *minus, const int32_t *lab, PackError *err)
{
  FILE *o = fopen(out_path, "wb");
  if (!o) { SetErrPublic(err, 0, std::string("Error: cannot create file '") + out_path + "'!", true); return false; }
  const uint32_t n_chrom = (uint32_t)names.size(), flags = lab ? 1u : 0u;
  size_t off = 0;
  auto put = [&](const void *p, size_t bytes) { if (bytes) fwrite(p, 1, bytes, o); off += bytes; };
  auto pad = [&]() { static const char z[8] = {0}; const size_t k = Pad8(off) - off; if (k) put(z, k); };
  put(kGtxMagic, 8); put(&n_chrom, 4); put(&flags, 4); put(&n, 8);
  for (const std::string &nm : names) { const uint16_t len = (uint16_t)nm.size(); put(&len, 2); put(nm.data(), len); }
  pad(); put(cidx, 2 * n); pad(); put(st, 4 * n); pad(); put(en, 4 * n); pad(); put(minus, (size_t)((n + 7) / 8)); pad();
  if (lab) { put(lab, 4 * n); pad(); }
  const bool ok = fclose(o) == 0;
  if (!ok) SetErrPublic(err, 0, std::string("Error: cannot write file '") + out_path + "'!", true);
  return ok;
}

// ---------------------------------------------------------------------------------------------------
// ChromTable
// ---------------------------------------------------------------------------------------------------
void ChromTable::Add(const char *name)
{
  for (const std::string &n : names_) if (n == name) return;
  names_.push_back(name);
  frozen_ = false;
}

void ChromTable::Freeze()
{
  std::sort(names_.begin(), names_.end(), [](const std::string &a, const std::string &b) { return strcmp(a.c_str(), b.c_str()) < 0; });
  frozen_ = true;
}

int ChromTable::Find(const char *name) const
{
  int lo = 0, hi = (int)names_.size();
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    int d = strcmp(names_[mid].c_str(), name);
    if (d == 0) return mid;
    if (d < 0) lo = mid + 1; else hi = mid;
  }
  return -1;
}

// ---------------------------------------------------------------------------------------------------
// BedPacker
// ---------------------------------------------------------------------------------------------------
namespace {

// (aligned to two cache lines: the order-check fields at its end are written for every line, the pointers at its start are read
//  for every line by the thread of the NEXT piece -- on one line they cost 6x in parse speed at 16 threads)
struct alignas(128) Piece {          // one thread's share of a block
  char *begin = nullptr, *end = nullptr;
  const GtxView *gtx = nullptr; uint64_t rec0 = 0, rec1 = 0;   // ... or of a packed file: records [rec0, rec1)
  long first_line = 0; long n_lines = 0;
  std::vector<int32_t> tri, w, zero_len;
  std::vector<int32_t> m_tri, m_w, m_cnt, m_blocks;   // collect_blocks (PackedBatch::m_*)
  std::vector<int32_t> tri_minus, w_minus;   // strand-aware runs: '-' reads are grouped behind the '+' reads of the batch
  // strand-blind runs write straight into the batch (room for one read per line was made there): no per-piece
  // buffer, no copy -- fresh memory is what packing costs (first-touch page faults do not run in parallel)
  int32_t *dtri = nullptr, *dw = nullptr; size_t nd = 0;
  PackError err;
  int64_t label_sum = 0;
  std::vector<long> cur_blocks;      // explode_blocks: the intervals (start, stop pairs) of the region being handled, empty = its one interval
  // order-check context of the regions in this piece
  bool any = false;
  const char *prev_chrom = nullptr; size_t prev_chrom_len = 0; int prev_id = -2;   // the line before (text still in the block) and its class lookup
  std::string first_chrom, last_chrom; char first_strand = '+', last_strand = '+';
  long first_start = 0, last_start = 0, first_region_line = 0;
};

// true when (chrom, [strand,] start) sorts before the previous region's key (genomic_intervals.cpp:396-401)
inline bool SortsBefore(const char *chrom, char strand, long start, const char *pchrom, char pstrand, long pstart, bool by_strand)
{
  int d = strcmp(chrom, pchrom);
  if (d) return d < 0;
  if (by_strand && strand != pstrand) return strand < pstrand;
  return start < pstart;
}

inline void SetErr(PackError *e, long line, const std::string &msg, bool no_prefix = false)
{
  if (!e->set) { e->set = true; e->line = line; e->msg = msg; e->no_prefix = no_prefix; }
}

const char *NotSortedMsg(const PackOptions &o)
{
  const bool overlaps = o.mode == PACK_OVERLAPS_SORTED;
  if (overlaps) return o.sorted_by_strand ? "query regions are not sorted (sorted-by-strand = true)!" : "query regions are not sorted (sorted-by-strand = false)!";
  return o.sorted_by_strand ? "input regions are not sorted (sorted-by-strand = true)!" : "input regions are not sorted (sorted-by-strand = false)!";
}

// one query against the pull loop of LoadIndexBuffer (genomic_intervals.cpp:5851-5870); false when it pulls region v
static inline bool GuardStep(IndexGuard *g, const BedFields &f)
{
  const long v = (long)g->start.size();
  while (g->p < v) {
    const long k = g->p;
    int d = strcmp(f.chrom, g->chrom[k]);                          // CalcDirection (:1225-1236)
    if (d == 0 && g->by_strand) d = (int)f.strand - (int)g->strand[k];
    if (d == 0) d = g->stop[k] < f.start ? 1 : (f.stop < g->start[k] ? -1 : 0);
    if (d < 0) break;
    if (++g->p == v) return false;                                 // Next() delivers region v: IsBefore(region v-1) holds
  }
  return true;
}

// What happens to one parsed region (the same for a text line and for a record of a packed file): order check,
// chromosome lookup, the mode's validity rules, output.  Returns false when an error was recorded.
// (sorted input repeats a chromosome millions of times in a row: the name of the line before is compared first -- by address for the
//  records of a packed file -- and its order key, class id and strcmp are reused)
static inline bool HandleRecord(Piece *p, const PackOptions &o, const BedFields &f, size_t chrom_len, long label_value, long line_no)
{
  const bool sorted_mode = o.mode == PACK_OVERLAPS_SORTED || o.mode == PACK_SCAN_SORTED;
  const bool weighted = o.max_label_value > 1;
  const int n_chrom = o.chroms->size();
    const bool same = p->prev_chrom && chrom_len == p->prev_chrom_len && (f.chrom == p->prev_chrom || memcmp(f.chrom, p->prev_chrom, chrom_len) == 0);
    if (sorted_mode) {
      if (p->any) {
        const bool before = same ? ((o.sorted_by_strand && f.strand != p->last_strand) ? f.strand < p->last_strand : f.start < p->last_start)
                                 : SortsBefore(f.chrom, f.strand, f.start, p->prev_chrom, p->last_strand, p->last_start, o.sorted_by_strand);
        if (before) { SetErr(&p->err, line_no, NotSortedMsg(o)); return false; }
      } else { p->first_chrom = f.chrom; p->first_strand = f.strand; p->first_start = f.start; p->first_region_line = line_no; }
      p->last_strand = f.strand; p->last_start = f.start; p->any = true;
    }
    if (o.guard && o.mode == PACK_OVERLAPS_SORTED && !GuardStep(o.guard, f)) { SetErr(&p->err, line_no, o.guard->msg, true); return false; }
    if (!same) { p->prev_id = o.chroms->Find(f.chrom); p->prev_chrom = f.chrom; p->prev_chrom_len = chrom_len; }
    const int id = p->prev_id;
    bool zero_len = false;
    long wv = 1;                                                         // GetLabelValue (genomic_intervals.cpp:1081-1085)
    if (weighted) { const long v = label_value; wv = v < o.max_label_value ? v : o.max_label_value; }
    p->label_sum += wv;
    switch (o.mode) {
      case PACK_OVERLAPS_UNSORTED:
        if (id < 0) return true;                                            // unknown chromosome: never validated
        if (f.stop <= 0) { SetErr(&p->err, line_no, "stop position must be positive!"); break; }
        if (f.start > f.stop) { SetErr(&p->err, line_no, "start position cannot be greater than stop position!"); break; }
        break;
      case PACK_OVERLAPS_SORTED:
        if (id < 0) return true;
        zero_len = f.start == f.stop + 1;                                  // (inverted reads go on: the library matches them pair by pair)
        break;
      case PACK_SCAN_UNSORTED:
        if (f.start > f.stop || f.stop <= 0) return true;
        if (id < 0) return true;
        break;
      case PACK_SCAN_SORTED:
        if (id < 0) return true;
        break;
    }
    if (p->err.set) return false;
    const bool minus = o.strand_aware && f.strand == '-';
    const int32_t cls = (int32_t)(id + (minus ? n_chrom : 0));
    if (!p->cur_blocks.empty() && o.collect_blocks) {
      p->m_tri.push_back(cls); p->m_tri.push_back((int32_t)f.start); p->m_tri.push_back((int32_t)f.stop);
      p->m_w.push_back((int32_t)wv); p->m_cnt.push_back((int32_t)(p->cur_blocks.size() / 2));
      for (long x : p->cur_blocks) p->m_blocks.push_back((int32_t)x);
    } else if (!p->cur_blocks.empty()) {                                 // (never in direct mode: PackPieces)
      std::vector<int32_t> &dst = minus ? p->tri_minus : p->tri;
      for (size_t b = 0; b + 1 < p->cur_blocks.size(); b += 2) {
        dst.push_back(cls); dst.push_back((int32_t)p->cur_blocks[b]); dst.push_back((int32_t)p->cur_blocks[b + 1]);
        if (weighted) (minus ? p->w_minus : p->w).push_back((int32_t)wv);
      }
    } else if (p->dtri) {
      int32_t *d = p->dtri + 3 * p->nd;
      d[0] = cls; d[1] = (int32_t)f.start; d[2] = (int32_t)f.stop;
      if (weighted) p->dw[p->nd] = (int32_t)wv;
      p->nd++;
    } else {
      std::vector<int32_t> &dst = minus ? p->tri_minus : p->tri;
      dst.push_back(cls); dst.push_back((int32_t)f.start); dst.push_back((int32_t)f.stop);
      if (weighted) (minus ? p->w_minus : p->w).push_back((int32_t)wv);
    }
    if (zero_len && o.collect_zero_length) { p->zero_len.push_back(cls); p->zero_len.push_back((int32_t)f.start); p->zero_len.push_back((int32_t)wv); }
  return true;
}

void ParsePiece(Piece *p, const PackOptions &o)
{
  const bool weighted = o.max_label_value > 1;
  long line_no = p->first_line - 1;
  if (p->gtx) {                                              // records of a packed file: already tokenised, numbers already read
    const GtxView &g = *p->gtx;
    if (!p->dtri) { p->tri.reserve((size_t)(p->rec1 - p->rec0) * 3); if (weighted) p->w.reserve((size_t)(p->rec1 - p->rec0)); }
    for (uint64_t r = p->rec0; r < p->rec1; r++) {
      line_no++;
      BedFields f;
      f.chrom = (char *)g.chrom[g.chrom_idx[r]].c_str(); f.label = nullptr;
      f.start = g.start[r]; f.stop = g.stop[r];
      f.strand = (g.minus[r >> 3] >> (r & 7)) & 1 ? '-' : '+';
      f.n_tokens = 6;
      if (!HandleRecord(p, o, f, g.chrom[g.chrom_idx[r]].size(), g.label ? g.label[r] : 0, line_no)) break;
    }
    if (p->prev_chrom) p->last_chrom = p->prev_chrom;
    p->n_lines = line_no - (p->first_line - 1);
    return;
  }
  char *cur = p->begin;
  size_t est = (size_t)(p->end - p->begin) / 20 + 16;
  if (!p->dtri) { p->tri.reserve(est * 3); if (weighted) p->w.reserve(est); }
  while (cur < p->end) {
    BedFields f; char *bad = nullptr;
    BedStatus st = BED_OK;
    static const bool fast_lines = getenv("GTX_NO_FAST_PARSE") == nullptr;      // (the tests compare both ways)
    size_t chrom_len = 0;
    char *next = fast_lines ? ParseTabbedLine(cur, p->end, &f, &chrom_len) : nullptr;
    if (next) { cur = next; line_no++; }
    else {
      char *nl = (char *)memchr(cur, '\n', (size_t)(p->end - cur));
      if (!nl) break;
      *nl = 0;
      char *line = cur;
      cur = nl + 1;
      line_no++;
      st = ParseBedLine(line, &f, &bad);
      if (st == BED_OK) chrom_len = strlen(f.chrom);
    }
    if (st == BED_TOO_FEW_TOKENS) { SetErr(&p->err, line_no, "number of tokens should be at least 3 for BED format!"); break; }
    if (st == BED_BAD_STRAND) { SetErr(&p->err, line_no, std::string("Error: invalid strand '") + bad + "'!", true); break; }
    const long label_value = f.label ? FastAtol(f.label) : 0;      // (before the block lists are cut into tokens)
    if (f.n_tokens == 12) {
      // a multi-interval region: under -gaps it is matched on its envelope [first interval's start, last interval's stop]
      // (genomic_intervals.cpp:5226, :5752, :5278); its intervals must be sorted and disjoint (:1153-1161, checked at :5709, :5880)
      const bool overlaps = o.mode == PACK_OVERLAPS_SORTED || o.mode == PACK_OVERLAPS_UNSORTED;
      if (!overlaps || !(o.match_gaps || o.explode_blocks || o.collect_blocks) || f.n_blocks < 1) { SetErr(&p->err, line_no, "multi-interval (BED12) regions are outside the MI355X counting path (except genomic_overlaps count, coverage and density)!"); break; }
      std::vector<long> iv; BedBlocks(f, &iv);
      bool ok = true;
      for (size_t k = 2; k < iv.size(); k += 2) if (iv[k] < iv[k - 2] || iv[k] <= iv[k - 1]) ok = false;
      if (!ok) { SetErr(&p->err, line_no, "query regions should be compatible, sorted and non-overlapping!"); break; }
      if (o.collect_blocks) for (size_t k = 3; k < iv.size(); k += 2) if (iv[k] < iv[k - 2]) ok = false;     // (a block of negative size)
      if (!ok) { SetErr(&p->err, line_no, "multi-interval (BED12) region with an interval of negative size is outside the MI355X counting path!"); break; }
      f.start = iv.front(); f.stop = iv.back();
      if ((o.explode_blocks || o.collect_blocks) && iv.size() > 2) {
        for (long x : iv) if (x >= INT_MAX - 1 || x <= INT_MIN + 1) { SetErr(&p->err, line_no, "coordinate does not fit the packed 32-bit representation of the MI355X path!"); break; }
        if (p->err.set) break;
        p->cur_blocks.swap(iv);
      }
    }
    if (f.start >= INT_MAX - 1 || f.stop >= INT_MAX - 1 || f.start <= INT_MIN + 1 || f.stop <= INT_MIN + 1) {
      SetErr(&p->err, line_no, "coordinate does not fit the packed 32-bit representation of the MI355X path!"); break;
    }
    const bool more = HandleRecord(p, o, f, chrom_len, label_value, line_no);
    p->cur_blocks.clear();
    if (!more) break;
  }
  if (p->prev_chrom) p->last_chrom = p->prev_chrom;            // (a copy: the seam check of the next block outlives this one's text)
  p->n_lines = line_no - (p->first_line - 1);
}

long CountLines(const char *b, const char *e)
{
  long n = 0;
  const __m128i vn = _mm_set1_epi8('\n');
  for (; e - b >= 64; b += 64) {                              // 64 bytes per turn: four compares, one population count
    const unsigned long long m = (unsigned long long)(unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)b), vn)) |
                                 ((unsigned long long)(unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(b + 16)), vn)) << 16) |
                                 ((unsigned long long)(unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(b + 32)), vn)) << 32) |
                                 ((unsigned long long)(unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(b + 48)), vn)) << 48);
    n += __builtin_popcountll(m);
  }
  for (; b < e; b++) n += *b == '\n';
  return n;
}

}  // namespace

// what the bulk readers of this library use by default: GTX_PACK_THREADS, else the container's CPU quota (<= 64)
int WorkerThreads()
{
  const char *e = getenv("GTX_PACK_THREADS");
  if (e && atoi(e) > 0) return atoi(e);
  return std::min(EffectiveCpus(), 64);
}

long CountNewlines(const char *b, const char *e) { return CountLines(b, e); }

void ParallelFor(int n, const std::function<void(int)> &fn) { Pool::Run(n, fn); }

BedPacker::BedPacker(LineSource *src, const PackOptions &opt) : src_(src), opt_(opt)
{
  if (opt_.guard) opt_.threads = 1;
  if (opt_.threads <= 0) { const char *e = getenv("GTX_PACK_THREADS"); if (e && atoi(e) > 0) opt_.threads = atoi(e); }
  if (opt_.threads <= 0) opt_.threads = std::min(EffectiveCpus(), 64);
}

BedPacker::BedPacker(const GtxView *packed, const PackOptions &opt) : src_(nullptr), opt_(opt)
{
  if (opt_.guard) opt_.threads = 1;
  if (opt_.threads <= 0) { const char *e = getenv("GTX_PACK_THREADS"); if (e && atoi(e) > 0) opt_.threads = atoi(e); }
  if (opt_.threads <= 0) opt_.threads = std::min(EffectiveCpus(), 64);
  gtx_ = packed;
}

void BedPacker::Prime(const std::string &line, long line_no)
{
  primed_.assign(line.begin(), line.end()); primed_.push_back('\n');
  primed_first_line_ = line_no; primed_set_ = true;
}

void BedPacker::PrimeBlock(const std::string &lines, long first_line)
{
  primed_.assign(lines.begin(), lines.end());
  primed_first_line_ = first_line; primed_set_ = !lines.empty();
}

// parse one block of complete lines with the thread pool and append the result to *out
bool BedPacker::PackBlock(char *block, size_t got, long first_line, PackedBatch *out, PackError *err)
{
  // cut the block into pieces at line ends
  int T = (int)std::min<size_t>((size_t)opt_.threads, got / (256u << 10) + 1);
  std::vector<Piece> pieces(T);
  char *b = block, *e = block + got;
  for (int t = 0; t < T; t++) {
    char *pe = t == T - 1 ? e : b + (size_t)(e - b) / (size_t)(T - t);
    if (t != T - 1) { char *nl = (char *)memchr(pe, '\n', (size_t)(e - pe)); pe = nl ? nl + 1 : e; }
    pieces[t].begin = b; pieces[t].end = pe; b = pe;
  }
  // line numbers: count in parallel, prefix, then parse in parallel
  const double tc = NowMs();
  Pool::Run(T, [&pieces](int t) { pieces[t].n_lines = CountLines(pieces[t].begin, pieces[t].end); });
  if (kTrace) fprintf(stderr, "[pack]   lines counted in %.1f ms (%d pieces)\n", NowMs() - tc, T);
  return PackPieces(&pieces, first_line, out, err);
}

// pieces with their line counts known -> parsed in parallel, checked at the seams, appended to *out
bool BedPacker::PackPieces(void *pieces_ptr, long first_line, PackedBatch *out, PackError *err)
{
  std::vector<Piece> &pieces = *(std::vector<Piece> *)pieces_ptr;
  const int T = (int)pieces.size();
  const bool sorted_mode = opt_.mode == PACK_OVERLAPS_SORTED || opt_.mode == PACK_SCAN_SORTED;
  long ln = first_line;
  for (int t = 0; t < T; t++) { pieces[t].first_line = ln; ln += pieces[t].n_lines; }
  const bool direct = !opt_.strand_aware && !opt_.explode_blocks;   // (one read per line is what the direct mode makes room for)
  const bool weighted = opt_.max_label_value > 1;
  const size_t base_tri = out->tri.size(), base_w = out->w.size();
  if (direct) {
    const size_t total = (size_t)(ln - first_line);
    out->tri.resize(base_tri + 3 * total);
    if (weighted) out->w.resize(base_w + total);
    size_t before = 0;
    for (int t = 0; t < T; t++) {
      pieces[t].dtri = out->tri.data() + base_tri + 3 * before;
      pieces[t].dw = weighted ? out->w.data() + base_w + before : nullptr;
      before += (size_t)pieces[t].n_lines;
    }
  }
  const double tp = NowMs();
  Pool::Run(T, [&pieces, this](int t) { ParsePiece(&pieces[t], opt_); });
  if (kTrace) fprintf(stderr, "[pack]   parsed in %.1f ms\n", NowMs() - tp);
  // seams and errors in file order; the first error in file order wins
  for (int t = 0; t < T; t++) {
    Piece &p = pieces[t];
    if (sorted_mode && p.any && have_prev_ &&
        SortsBefore(p.first_chrom.c_str(), p.first_strand, p.first_start, prev_chrom_.c_str(), prev_strand_, prev_start_, opt_.sorted_by_strand)) {
      if (!p.err.set || p.err.line > p.first_region_line) { p.err = PackError(); SetErr(&p.err, p.first_region_line, NotSortedMsg(opt_)); }
    }
    if (p.err.set) {
      *err = p.err;
      if (!opt_.keep_prefix_on_error) { if (direct) { out->tri.resize(base_tri); out->w.resize(base_w); } return false; }
      // the regions in front of the offending line stay: the pieces before this one, and what this piece had parsed when it met the
      // line (nothing of it when the line is its first region: the order error of a seam)
      const bool own = p.any && p.err.line > p.first_region_line;     // (p.any: a region of this piece came before the error)
      if (own) { have_prev_ = true; prev_chrom_ = p.prev_chrom ? std::string(p.prev_chrom, p.prev_chrom_len) : p.last_chrom; prev_strand_ = p.last_strand; prev_start_ = p.last_start; }
      err->have_last = have_prev_; err->last_chrom = prev_chrom_; err->last_strand = prev_strand_; err->last_start = prev_start_;
      const int keep = own ? t + 1 : t;
      if (direct) {
        size_t wpos = 0;
        for (int q = 0; q < keep; q++) {
          if (pieces[q].nd && out->tri.data() + base_tri + 3 * wpos != pieces[q].dtri) {
            memmove(out->tri.data() + base_tri + 3 * wpos, pieces[q].dtri, pieces[q].nd * 3 * sizeof(int32_t));
            if (weighted) memmove(out->w.data() + base_w + wpos, pieces[q].dw, pieces[q].nd * sizeof(int32_t));
          }
          wpos += pieces[q].nd;
        }
        out->tri.resize(base_tri + 3 * wpos);
        if (weighted) out->w.resize(base_w + wpos);
      } else {
        for (int pass = 0; pass < 2; pass++)
          for (int q = 0; q < keep; q++) {
            const std::vector<int32_t> &tr = pass ? pieces[q].tri_minus : pieces[q].tri, &ww = pass ? pieces[q].w_minus : pieces[q].w;
            out->tri.insert(out->tri.end(), tr.begin(), tr.end()); out->w.insert(out->w.end(), ww.begin(), ww.end());
          }
      }
      for (int q = 0; q < keep; q++) out->label_sum += pieces[q].label_sum;
      return false;
    }
    if (sorted_mode && p.any) { have_prev_ = true; prev_chrom_ = p.last_chrom; prev_strand_ = p.last_strand; prev_start_ = p.last_start; }
  }
  if (direct) {
    // close the gaps that dropped lines left (unknown chromosomes, reads outside the rules): usually there are none
    size_t wpos = 0, before = 0;
    for (int t = 0; t < T; t++) {
      if (wpos != before && pieces[t].nd) {
        memmove(out->tri.data() + base_tri + 3 * wpos, pieces[t].dtri, pieces[t].nd * 3 * sizeof(int32_t));
        if (weighted) memmove(out->w.data() + base_w + wpos, pieces[t].dw, pieces[t].nd * sizeof(int32_t));
      }
      wpos += pieces[t].nd; before += (size_t)pieces[t].n_lines;
    }
    out->tri.resize(base_tri + 3 * wpos);
    if (weighted) out->w.resize(base_w + wpos);
    for (int t = 0; t < T; t++) {
      out->zero_len.insert(out->zero_len.end(), pieces[t].zero_len.begin(), pieces[t].zero_len.end());
      out->m_tri.insert(out->m_tri.end(), pieces[t].m_tri.begin(), pieces[t].m_tri.end()); out->m_w.insert(out->m_w.end(), pieces[t].m_w.begin(), pieces[t].m_w.end());
      out->m_cnt.insert(out->m_cnt.end(), pieces[t].m_cnt.begin(), pieces[t].m_cnt.end()); out->m_blocks.insert(out->m_blocks.end(), pieces[t].m_blocks.begin(), pieces[t].m_blocks.end());
      out->n_lines += pieces[t].n_lines;
      out->label_sum += pieces[t].label_sum;
    }
    return true;
  }
  // concatenate the pieces (each thread copies its own piece to its final place): all '+' parts in
  // file order, then all '-' parts in file order -- counting does not depend on the order of the
  // reads, and a position-sorted strand-aware stream becomes two class-sorted runs for the kernel
  std::vector<size_t> at_tri(2 * T + 1), at_w(2 * T + 1);
  at_tri[0] = out->tri.size(); at_w[0] = out->w.size();
  for (int t = 0; t < T; t++) { at_tri[t + 1] = at_tri[t] + pieces[t].tri.size(); at_w[t + 1] = at_w[t] + pieces[t].w.size(); }
  for (int t = 0; t < T; t++) { at_tri[T + t + 1] = at_tri[T + t] + pieces[t].tri_minus.size(); at_w[T + t + 1] = at_w[T + t] + pieces[t].w_minus.size(); }
  out->tri.resize(at_tri[2 * T]); out->w.resize(at_w[2 * T]);
  {
    auto copy_piece = [&](int t) {
      if (!pieces[t].tri.empty()) memcpy(out->tri.data() + at_tri[t], pieces[t].tri.data(), pieces[t].tri.size() * sizeof(int32_t));
      if (!pieces[t].w.empty()) memcpy(out->w.data() + at_w[t], pieces[t].w.data(), pieces[t].w.size() * sizeof(int32_t));
      if (!pieces[t].tri_minus.empty()) memcpy(out->tri.data() + at_tri[T + t], pieces[t].tri_minus.data(), pieces[t].tri_minus.size() * sizeof(int32_t));
      if (!pieces[t].w_minus.empty()) memcpy(out->w.data() + at_w[T + t], pieces[t].w_minus.data(), pieces[t].w_minus.size() * sizeof(int32_t));
    };
    Pool::Run(T, copy_piece);
  }
  for (int t = 0; t < T; t++) {
    out->zero_len.insert(out->zero_len.end(), pieces[t].zero_len.begin(), pieces[t].zero_len.end());
    out->m_tri.insert(out->m_tri.end(), pieces[t].m_tri.begin(), pieces[t].m_tri.end()); out->m_w.insert(out->m_w.end(), pieces[t].m_w.begin(), pieces[t].m_w.end());
    out->m_cnt.insert(out->m_cnt.end(), pieces[t].m_cnt.begin(), pieces[t].m_cnt.end()); out->m_blocks.insert(out->m_blocks.end(), pieces[t].m_blocks.begin(), pieces[t].m_blocks.end());
    out->n_lines += pieces[t].n_lines;
    out->label_sum += pieces[t].label_sum;
  }
  return true;
}

bool BedPacker::NextBatch(PackedBatch *out, size_t target_reads, PackError *err)
{
  out->clear();    // (the batch object is the caller's and is reused)
  out->tri.reserve(target_reads * 3 + (24u << 20));       // one allocation; its pages are first touched by the copy threads
  if (opt_.max_label_value > 1) out->w.reserve(target_reads + (8u << 20));
  if (primed_set_) {
    primed_set_ = false;
    if (!PackBlock(primed_.data(), primed_.size(), primed_first_line_, out, err)) return false;
  }
  if (gtx_) {
    // records of a packed file: cut the next target_reads of them into one range per thread
    const uint64_t left = gtx_->n - gtx_pos_;
    if (left == 0) return false;
    const uint64_t take = std::min<uint64_t>(left, target_reads);
    const int T = (int)std::min<uint64_t>((uint64_t)opt_.threads, take / 65536 + 1);
    std::vector<Piece> pieces(T);
    for (int t = 0; t < T; t++) {
      pieces[t].gtx = gtx_;
      pieces[t].rec0 = gtx_pos_ + take * (uint64_t)t / T; pieces[t].rec1 = gtx_pos_ + take * (uint64_t)(t + 1) / T;
      pieces[t].n_lines = (long)(pieces[t].rec1 - pieces[t].rec0);
    }
    const bool ok = PackPieces(&pieces, (long)gtx_pos_ + 1, out, err);
    gtx_pos_ += take;
    return ok && gtx_pos_ < gtx_->n;
  }
  if (!src_ || exhausted_) return false;
  // the next block is read (and inflated, for .gz) while the current one is parsed
  const size_t block_bytes = (getenv("GTX_PACK_BLOCK_MB") && atoi(getenv("GTX_PACK_BLOCK_MB")) > 0 ? (size_t)atoi(getenv("GTX_PACK_BLOCK_MB")) : 64u) << 20;
  auto read_block = [this, block_bytes](int buf) { Ahead a; long fl = 0; a.buf = buf; a.got = src_->NextBlockView(blocks_[buf], &a.view, block_bytes, &fl); return a; };
  if (!ahead_.valid()) { ahead_ = std::async(std::launch::async, read_block, next_buf_); next_buf_ ^= 1; }
  while (out->tri.size() / 3 < target_reads) {
    const double t0 = NowMs();
    Ahead cur = ahead_.get();
    if (kTrace) fprintf(stderr, "[pack] waited %.1f ms for a block of %zu bytes\n", NowMs() - t0, cur.got);
    if (cur.got == 0) { ahead_ = std::future<Ahead>(); exhausted_ = true; return false; }
    const long first_line = src_->line_no() + 1;
    ahead_ = std::async(std::launch::async, read_block, next_buf_); next_buf_ ^= 1;      // the other buffer: cur's is being parsed
    const int64_t before = out->n_lines;
    const double t1 = NowMs();
    bool ok = PackBlock(cur.view, cur.got, first_line, out, err);
    if (kTrace) fprintf(stderr, "[pack] block packed in %.1f ms\n", NowMs() - t1);
    src_->AdvanceLines((long)(out->n_lines - before));
    if (!ok) { ahead_.wait(); return false; }
  }
  return true;
}

bool BedPacker::PackPrimedText(PackedBatch *out, PackError *err)
{
  out->clear();
  if (!primed_set_) return true;
  primed_set_ = false;
  const bool ok = PackBlock(primed_.data(), primed_.size(), primed_first_line_, out, err);
  seam_have_ = have_prev_; seam_chrom_ = prev_chrom_; seam_strand_ = prev_strand_; seam_start_ = prev_start_;
  return ok;
}

bool BedPacker::NextTextBlock(TextBlock *b)
{
  if (!src_ || exhausted_) return false;
  const size_t block_bytes = (getenv("GTX_PACK_BLOCK_MB") && atoi(getenv("GTX_PACK_BLOCK_MB")) > 0 ? (size_t)atoi(getenv("GTX_PACK_BLOCK_MB")) : 64u) << 20;
  auto read_block = [this, block_bytes](int buf) { Ahead a; long fl = 0; a.buf = buf; a.got = src_->NextBlockView(blocks_[buf], &a.view, block_bytes, &fl); return a; };
  // (no read-ahead into the other buffer here: the caller may still need the block before this one -- it reads while the device works)
  Ahead cur;
  if (text_buf_[0]) {                                               // the caller's page-locked buffers: read where the DMA engine reads
    long fl = 0;
    cur.buf = next_buf_; cur.view = text_buf_[next_buf_];
    cur.got = src_->ReadTextInto(cur.view, std::min(text_cap_, block_bytes), &fl);
    if (cur.got == (size_t)-1) { text_buf_[0] = text_buf_[1] = nullptr; cur = read_block(next_buf_); }   // (a line longer than the buffer, stdin, .gz)
  } else cur = read_block(next_buf_);
  next_buf_ ^= 1;
  if (cur.got == 0) { exhausted_ = true; return false; }
  b->text = cur.view; b->bytes = cur.got; b->first_line = src_->line_no() + 1;
  {
    const int T = (int)std::min<size_t>((size_t)opt_.threads, cur.got / (1u << 20) + 1);
    std::vector<long> part(T, 0);
    char *v = cur.view; const size_t got = cur.got;
    Pool::Run(T, [&](int t) { part[t] = CountLines(v + got * (size_t)t / T, v + got * (size_t)(t + 1) / T); });
    long n = 0; for (long x : part) n += x;
    b->n_lines = n;
  }
  src_->AdvanceLines((long)b->n_lines);
  b->have_prev = seam_have_; b->prev_chrom = seam_chrom_; b->prev_strand = seam_strand_; b->prev_start = seam_start_; b->seam_ok = seam_ok_;
  // the order key of the block's last line (a copy of it is parsed: the block itself stays as it is)
  if (b->bytes >= 2) {
    const char *e = b->text + b->bytes - 1;                        // the final newline
    const char *s = e;
    while (s > b->text && s[-1] != '\n') s--;
    std::string line(s, e);
    BedFields f; char *bad = nullptr;
    if (!line.empty() && line.find('\0') == std::string::npos && ParseBedLine(&line[0], &f, &bad) == BED_OK && f.n_tokens != 12) {
      seam_have_ = true; seam_chrom_ = f.chrom; seam_strand_ = f.strand; seam_start_ = f.start;
    } else seam_ok_ = false;
  }
  return true;
}

bool BedPacker::PackTextBlock(const TextBlock &b, PackedBatch *out, PackError *err)
{
  have_prev_ = b.have_prev; prev_chrom_ = b.prev_chrom; prev_strand_ = b.prev_strand; prev_start_ = b.prev_start;
  return PackBlock(b.text, b.bytes, b.first_line, out, err);
}

}  // namespace gtxhost
