// gtx_text.h -- launch interface of the device-side BED tokenizer (gtx_text.hip), used by gtx_capi.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "gtx.h"

namespace gtxtext {

struct TextDevice {               // device buffers of one block in flight
  const char *text;               // the block's bytes
  unsigned *segCount;             // [segments + 1] newlines per 1 KB segment, then their exclusive prefix
  unsigned *nl;                   // [n_lines] byte offset of every newline
  int *tri; int *w;               // [n_lines] packed triples / weights
  int *tri2; int *w2; unsigned *blkMinus;   // strand-aware runs: the same grouped by strand ('+' first), and the '-' lines per 128-line block [n_lines / 128 + 2]; null otherwise
  int *flag;                      // != 0: the block is not plain (nothing of it is counted)
  unsigned long long *blockSum, *labelSum;   // scans (may be null): the label values of the block's lines, added to the call's total when the block is plain
};
struct TextTables {               // per reference set: hash table of the chromosome names, the names, the seam's name behind them
  const void *table; unsigned tableMask; const char *names; unsigned prevOff, prevLen;
};
// scanRules: 0 = the overlap algorithms' rules (gtx_text_rules::sorted_rules), 1 = the unsorted scanner's (an interval with start > stop or
// stop <= 0 is skipped silently, genomic_intervals.cpp:5039), 2 = the sorted scanner's (nothing about the interval is checked; order as sorted_rules)
hipError_t launch_tokenize(const TextDevice &d, const TextTables &t, const gtx_text_rules &r, size_t bytes, unsigned nLines, hipStream_t st, int scanRules = 0);
void build_tables(const gtx_text_rules &r, std::vector<int32_t> *table, unsigned *mask, std::string *blob);

}  // namespace gtxtext
