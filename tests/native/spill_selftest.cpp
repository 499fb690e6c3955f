// Self-test of the trajectory's disk tier (pnode_amd/csrc/pn_spill.cpp) in host mode (device = 0: plain memory, no HIP call
// is executed), meant to run under -fsanitize=address,undefined and under -fsanitize=thread on the CPU: staging buffers,
// I/O thread, read-ahead, out-of-order requests, overwrite of a checkpoint, drop, statistics, clean-up.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "pnode_amd.h"

#define REQUIRE(c)                                                          \
  do {                                                                      \
    if (!(c)) {                                                             \
      std::fprintf(stderr, "%s:%d: REQUIRE(%s) failed: %s\n", __FILE__, __LINE__, #c, pn_last_error()); \
      std::exit(1);                                                         \
    }                                                                       \
  } while (0)

static void fill(std::vector<unsigned char> &b, int64_t id, int gen) {
  for (size_t i = 0; i < b.size(); ++i) b[i] = (unsigned char)((id * 131 + gen * 17 + i * 7) & 0xff);
}

int main(int argc, char **argv) {
  REQUIRE(argc == 2);
  const std::string dir = std::string(argv[1]) + "/ckpt";
  const int64_t bytes = 64 * 1024 + 24;
  const int n = 40;
  REQUIRE(pn_spill_create(nullptr, bytes, 4, 0, 0) == nullptr);
  REQUIRE(pn_spill_create(dir.c_str(), bytes, 1, 0, 0) == nullptr);       // needs >= 2 staging buffers
  pn_spill *sp = pn_spill_create(dir.c_str(), bytes, 3, 0, 0);
  REQUIRE(sp != nullptr);
  std::vector<unsigned char> src(bytes), dst(bytes), want(bytes);
  // forward sweep: more checkpoints than staging buffers, back to back
  for (int id = 0; id < n; ++id) {
    fill(src, id, 0);
    REQUIRE(pn_spill_put(sp, nullptr, id, src.data()) == 0);
  }
  // a checkpoint written twice keeps the newer contents
  fill(src, 7, 1);
  REQUIRE(pn_spill_put(sp, nullptr, 7, src.data()) == 0);
  // reverse sweep with read-ahead
  for (int id = n - 1; id >= 0; --id) {
    if (id > 0) REQUIRE(pn_spill_prefetch(sp, id - 1) == 0);
    REQUIRE(pn_spill_get(sp, nullptr, id, dst.data()) == 0);
    fill(want, id, id == 7 ? 1 : 0);
    REQUIRE(std::memcmp(dst.data(), want.data(), (size_t)bytes) == 0);
  }
  // out-of-order requests, repeated reads, prefetches that are never consumed
  const int order[] = {5, 31, 5, 0, 39, 12, 12, 7};
  for (int id : order) {
    REQUIRE(pn_spill_prefetch(sp, (id + 3) % n) == 0);
    REQUIRE(pn_spill_get(sp, nullptr, id, dst.data()) == 0);
    fill(want, id, id == 7 ? 1 : 0);
    REQUIRE(std::memcmp(dst.data(), want.data(), (size_t)bytes) == 0);
  }
  // a checkpoint rewritten while a read-ahead of its OLD contents may still be in flight: the later get
  // must return the new contents, never a torn or stale buffer
  int64_t extra_writes = 0;
  for (int rep = 0; rep < 25; ++rep) {
    const int id = 20 + (rep % 3);
    REQUIRE(pn_spill_prefetch(sp, id) == 0);
    fill(src, id, 2 + rep);
    REQUIRE(pn_spill_put(sp, nullptr, id, src.data()) == 0);
    ++extra_writes;
    REQUIRE(pn_spill_get(sp, nullptr, id, dst.data()) == 0);
    REQUIRE(std::memcmp(dst.data(), src.data(), (size_t)bytes) == 0);
  }
  REQUIRE(pn_spill_get(sp, nullptr, 1000, dst.data()) != 0);            // never written
  int64_t files = 0, bw = 0, br = 0, waits = 0;
  REQUIRE(pn_spill_stats(sp, &files, &bw, &br, &waits) == 0);
  REQUIRE(files == n && bw == (n + 1 + extra_writes) * bytes && br >= (n + 8) * bytes);
  REQUIRE(pn_spill_drop(sp, 3) == 0);
  REQUIRE(pn_spill_stats(sp, &files, nullptr, nullptr, nullptr) == 0 && files == n - 1);
  REQUIRE(pn_spill_get(sp, nullptr, 3, dst.data()) != 0);
  struct stat st;
  REQUIRE(stat((dir + "/SA-000004.bin").c_str(), &st) == 0 && st.st_size == bytes);
  REQUIRE(stat((dir + "/SA-000003.bin").c_str(), &st) != 0);
  pn_spill_destroy(sp);
  REQUIRE(stat(dir.c_str(), &st) != 0);                                 // files and directory removed
  // keep_files
  sp = pn_spill_create(dir.c_str(), bytes, 2, 0, 1);
  REQUIRE(sp != nullptr);
  fill(src, 1, 0);
  REQUIRE(pn_spill_put(sp, nullptr, 1, src.data()) == 0);
  pn_spill_destroy(sp);
  REQUIRE(stat((dir + "/SA-000001.bin").c_str(), &st) == 0);
  std::printf("spill selftest ok\n");
  return 0;
}
