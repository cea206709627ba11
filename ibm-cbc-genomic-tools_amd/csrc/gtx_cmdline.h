// gtx_cmdline.h -- "TOOL OPERATION [OPTIONS] FILES" parsing with the reference's rules
// (gtools/core.cpp:2420-2436): options are looked up by exact name, flags take no value, every
// other option consumes the next argument, parsing stops at the first token that does not start
// with '-', and an unknown option is an error.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace gtxhost {

class Options {
 public:
  void Flag(const char *name, bool *dst, const char *help) { *dst = false; items_.push_back({name, help, dst, nullptr, nullptr, nullptr, nullptr, nullptr}); }
  void Long(const char *name, long *dst, long def, const char *help) { *dst = def; items_.push_back({name, help, nullptr, dst, nullptr, nullptr, nullptr, nullptr}); }
  void ULong(const char *name, unsigned long *dst, unsigned long def, const char *help) { *dst = def; items_.push_back({name, help, nullptr, nullptr, dst, nullptr, nullptr, nullptr}); }
  void Double(const char *name, double *dst, double def, const char *help) { *dst = def; items_.push_back({name, help, nullptr, nullptr, nullptr, dst, nullptr, nullptr}); }
  void Str(const char *name, const char **dst, const char *def, const char *help) { *dst = def; items_.push_back({name, help, nullptr, nullptr, nullptr, nullptr, dst, nullptr}); }
  void Char(const char *name, char *dst, char def, const char *help) { *dst = def; items_.push_back({name, help, nullptr, nullptr, nullptr, nullptr, nullptr, dst}); }

  // argv[first..argc): returns the index of the first non-option argument
  int Parse(int argc, char **argv, int first)
  {
    int a = first;
    while (a < argc && argv[a][0] == '-') {
      Item *it = nullptr;
      for (Item &i : items_) if (!strcmp(i.name, argv[a])) { it = &i; break; }
      if (!it) { fprintf(stderr, "Error: unknown option '%s'!\n", argv[a]); exit(1); }
      if (it->flag) { *it->flag = true; a++; continue; }
      if (a + 1 >= argc) { fprintf(stderr, "Error: could not set option '%s'!\n", it->name); exit(1); }
      const char *v = argv[a + 1];
      if (it->l) *it->l = atol(v);
      else if (it->ul) *it->ul = strtoul(v, NULL, 10);
      else if (it->d) *it->d = atof(v);
      else if (it->s) *it->s = v;
      else if (it->c) *it->c = v[0];
      a += 2;
    }
    return a;
  }

  void Usage(const char *program, const char *operation, const char *args) const
  {
    fprintf(stderr, "\nUSAGE: \n  %s %s %s\n\nOPTIONS: \n", program, operation, args);
    for (const Item &i : items_) fprintf(stderr, "  %-25s %s\n", i.name, i.help);
    fprintf(stderr, "\n");
  }

 private:
  struct Item { const char *name, *help; bool *flag; long *l; unsigned long *ul; double *d; const char **s; char *c; };
  std::vector<Item> items_;
};

}  // namespace gtxhost
