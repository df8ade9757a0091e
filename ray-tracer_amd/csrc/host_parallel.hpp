// host_parallel.hpp -- fn(first, last) over [0, n) on the host threads this process may use (internal to libmi355rt).
// For the flat passes over a mesh of tens of millions of elements that precede the GPU's work: PLY body conversion,
// index validation, bounds.  Results must not depend on the split: callers combine per-range results with an
// associative rule.
#pragma once
#include <algorithm>
#include <cstddef>
#include <thread>
#include <vector>

#include <sched.h>

namespace mi355rt {

inline size_t host_threads_for(size_t n, size_t grain)
{
   int cores = 1;
   cpu_set_t set;
   if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = CPU_COUNT(&set);
   const size_t by_size = n / (grain ? grain : 1) + 1;
   return std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t) std::max(cores, 1), 16), by_size));
}

// fn(t, first, last): t = index of the range, 0 .. ranges - 1 (ranges = host_threads_for(n, grain))
template <class F>
void in_ranges(size_t n, size_t grain, F fn)
{
   const size_t nt = host_threads_for(n, grain);
   if (nt == 1) {
      fn((size_t) 0, (size_t) 0, n);
      return;
   }
   std::vector<std::thread> th;
   th.reserve(nt);
   for (size_t t = 0; t < nt; t++) th.emplace_back(fn, t, n * t / nt, n * (t + 1) / nt);
   for (std::thread &x : th) x.join();
}

}   // namespace mi355rt
