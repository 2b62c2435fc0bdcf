// kmdb_internal.h — shared by the translation units of libkmdb_amd.so (not installed).
#pragma once
#include <cstddef>
#include <string>
#include <utility>
#include <vector>

// records the message for kmdb_last_error() and returns a non-zero status
int kmdb_set_error(const std::string& msg);

// Gives the pages inside the regions back to the kernel, on up to `threads` threads (host_db.cpp).  madvise(MADV_DONTNEED) takes the
// address-space lock SHARED: the threads, and the page faults and allocations of every other thread, go on side by side, where a munmap
// of gigabytes holds the lock exclusively for as long as it frees pages.  The regions stay mapped (they read as zeros afterwards):
// unmapping them later, or the end of the process, finds nothing left to free.
void kmdb_drop_pages(const std::vector<std::pair<void*, size_t>>& regions, unsigned threads);
