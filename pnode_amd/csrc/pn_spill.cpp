// pnode_amd -- disk tier of the adjoint trajectory (include/pnode_amd.h section 5).
//
// PETSc's default TSTrajectory type ("basic") writes every checkpoint to a file of its own under
// -ts_trajectory_dirname and reads it back in the reverse sweep; the reference relies on it unless
// `-ts_trajectory_type memory` is given ("By default, disk is used", examples-pnode/ode_demo_petsc.py:26).
// Here HBM is the default tier; this engine is what `-ts_trajectory_type basic` selects.
//
// Design: checkpoints leave the device asynchronously.  pn_spill_put enqueues a device-to-host copy of the
// slot into one of `nbuf` pinned staging buffers on the CALLER'S stream (so it is ordered behind the kernels
// that wrote the slot and in front of the kernels that will recycle it) and hands the file write to an I/O
// thread, which waits for the copy's event first.  The forward sweep never waits for the file system unless
// every staging buffer is busy.  In the reverse sweep pn_spill_prefetch lets the I/O thread read the next
// checkpoint while the current step is being reversed; pn_spill_get waits for that read and enqueues the
// host-to-device copy.  One file per checkpoint: <dir>/SA-%06lld.bin, raw slot bytes.
#include <hip/hip_runtime_api.h>

#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "pnode_amd.h"
#include "pn_internal.h"

namespace {

enum BufState { FREE = 0, D2H_PENDING, WRITING, READING, READY, H2D_PENDING };

struct Buf {
  void *host = nullptr;
  hipEvent_t ev = nullptr;
  BufState state = FREE;
  int64_t id = -1;
};

struct Job {
  int kind;       // 0 = write buffer to file, 1 = read file into buffer
  int buf;
  int64_t id;
};

}  // namespace

struct pn_spill {
  std::string dir;
  int64_t slot_bytes = 0;
  bool device = true;
  bool keep_files = false;
  bool made_dir = false;
  std::vector<Buf> bufs;
  std::mutex mu;
  std::condition_variable cv_job, cv_buf;
  std::deque<Job> jobs;
  std::set<int64_t> files;
  std::vector<std::thread> workers;          // nbuf / 2 I/O threads: a single writer tops out at one memcpy into the page cache
  bool stop = false;
  std::string io_error;
  int64_t bytes_written = 0, bytes_read = 0, put_waits = 0, get_waits = 0;

  std::string path(int64_t id) const {
    char name[64];
    std::snprintf(name, sizeof name, "/SA-%06lld.bin", (long long)id);
    return dir + name;
  }

  void run() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
        if (jobs.empty()) return;          // stop requested and nothing left to do
        j = jobs.front();
        jobs.pop_front();
      }
      Buf &b = bufs[j.buf];
      std::string err;
      if (j.kind == 0) {
        if (device && hipEventSynchronize(b.ev) != hipSuccess) err = "device-to-host copy of a checkpoint failed";
        if (err.empty()) {
          FILE *f = std::fopen(path(j.id).c_str(), "wb");
          if (!f || std::fwrite(b.host, 1, (size_t)slot_bytes, f) != (size_t)slot_bytes)
            err = "cannot write " + path(j.id) + ": " + std::strerror(errno);
          if (f && std::fclose(f) != 0 && err.empty()) err = "cannot close " + path(j.id) + ": " + std::strerror(errno);
        }
        std::lock_guard<std::mutex> lk(mu);
        if (err.empty()) { files.insert(j.id); bytes_written += slot_bytes; }
        else if (io_error.empty()) io_error = err;
        b.state = FREE;
        b.id = -1;
      } else {
        FILE *f = std::fopen(path(j.id).c_str(), "rb");
        if (!f || std::fread(b.host, 1, (size_t)slot_bytes, f) != (size_t)slot_bytes)
          err = "cannot read " + path(j.id) + ": " + std::strerror(errno);
        if (f) std::fclose(f);
        std::lock_guard<std::mutex> lk(mu);
        if (err.empty()) bytes_read += slot_bytes;
        else if (io_error.empty()) io_error = err;
        b.state = READY;
      }
      cv_buf.notify_all();
    }
  }

  // a staging buffer nobody uses (mu held); buffers whose host-to-device copy has completed are reclaimed, and a
  // checkpoint that was read ahead but never asked for gives its buffer up (it is read again if it is wanted after all)
  int take_free(std::unique_lock<std::mutex> &lk, int64_t *waits) {
    bool waited = false;
    for (;;) {
      for (size_t i = 0; i < bufs.size(); ++i) {
        Buf &b = bufs[i];
        if (b.state == H2D_PENDING && (!device || hipEventQuery(b.ev) == hipSuccess)) { b.state = FREE; b.id = -1; }
        if (b.state == FREE) return (int)i;
      }
      for (size_t i = 0; i < bufs.size(); ++i)
        if (bufs[i].state == READY) { bufs[i].state = FREE; bufs[i].id = -1; return (int)i; }
      // everything in flight: wait for the I/O thread, or for the oldest host-to-device copy
      bool h2d = false;
      for (Buf &b : bufs) h2d = h2d || b.state == H2D_PENDING;
      if (!waited) { waited = true; ++*waits; }
      if (h2d) {
        for (Buf &b : bufs)
          if (b.state == H2D_PENDING) {
            hipEvent_t ev = b.ev;
            lk.unlock();
            (void)hipEventSynchronize(ev);
            lk.lock();
            break;
          }
      } else {
        cv_buf.wait(lk);
      }
    }
  }
};

extern "C" {

pn_spill *pn_spill_create(const char *dir, int64_t slot_bytes, int nbuf, int device, int keep_files) {
  if (!dir || !*dir || slot_bytes <= 0 || nbuf < 2) {
    pn::fail("pn_spill_create: need a directory, a positive slot size and at least 2 staging buffers");
    return nullptr;
  }
  pn_spill *sp = new pn_spill();
  sp->dir = dir;
  sp->slot_bytes = slot_bytes;
  sp->device = device != 0;
  sp->keep_files = keep_files != 0;
  struct stat st;
  if (stat(dir, &st) != 0) {
    if (mkdir(dir, 0777) != 0 && errno != EEXIST) {
      pn::fail(std::string("pn_spill_create: cannot create directory ") + dir + ": " + std::strerror(errno));
      delete sp;
      return nullptr;
    }
    sp->made_dir = true;
  } else if (!S_ISDIR(st.st_mode)) {
    pn::fail(std::string("pn_spill_create: ") + dir + " exists and is not a directory");
    delete sp;
    return nullptr;
  }
  sp->bufs.resize(nbuf);
  for (Buf &b : sp->bufs) {
    bool ok;
    if (sp->device) {
      ok = hipHostMalloc(&b.host, (size_t)slot_bytes, hipHostMallocDefault) == hipSuccess &&
           hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) == hipSuccess;
    } else {
      b.host = std::malloc((size_t)slot_bytes);
      ok = b.host != nullptr;
    }
    if (!ok) {
      pn::fail("pn_spill_create: cannot allocate the staging buffers");
      pn_spill_destroy(sp);
      return nullptr;
    }
  }
  const int nthreads = std::max(1, nbuf / 2);
  for (int i = 0; i < nthreads; ++i) sp->workers.emplace_back([sp] { sp->run(); });
  return sp;
}

void pn_spill_destroy(pn_spill *sp) {
  if (!sp) return;
  if (!sp->workers.empty()) {
    {
      std::lock_guard<std::mutex> lk(sp->mu);
      sp->stop = true;
    }
    sp->cv_job.notify_all();
    for (std::thread &w : sp->workers) w.join();
  }
  for (Buf &b : sp->bufs) {
    if (sp->device) {
      if (b.ev) { (void)hipEventSynchronize(b.ev); (void)hipEventDestroy(b.ev); }
      if (b.host) (void)hipHostFree(b.host);
    } else {
      std::free(b.host);
    }
  }
  if (!sp->keep_files) {
    for (int64_t id : sp->files) std::remove(sp->path(id).c_str());
    if (sp->made_dir) rmdir(sp->dir.c_str());      // only when empty
  }
  delete sp;
}

static int check_io(pn_spill *sp) {
  if (!sp->io_error.empty()) return pn::fail("trajectory disk tier: " + sp->io_error);
  return 0;
}

static void wait_written(pn_spill *sp, std::unique_lock<std::mutex> &lk, int64_t id);
static void wait_read(pn_spill *sp, std::unique_lock<std::mutex> &lk, int64_t id);

int pn_spill_put(pn_spill *sp, void *stream, int64_t id, const void *src) {
  std::unique_lock<std::mutex> lk(sp->mu);
  if (check_io(sp)) return 1;
  wait_written(sp, lk, id);          // an older copy of this checkpoint still on its way to the file: let it land first
  wait_read(sp, lk, id);             // a read-ahead of the OLD contents still in flight: it must not race the new write
                                     // on the same file (a torn read would sit in a READY buffer under this id)
  // a staging buffer that still holds an older copy of this checkpoint is stale now
  for (Buf &b : sp->bufs)
    if (b.id == id && (b.state == READY || b.state == H2D_PENDING)) { if (b.state == READY) { b.state = FREE; b.id = -1; } }
  const int i = sp->take_free(lk, &sp->put_waits);
  Buf &b = sp->bufs[i];
  b.state = D2H_PENDING;
  b.id = id;
  lk.unlock();
  if (sp->device) {
    hipError_t err = hipMemcpyAsync(b.host, src, (size_t)sp->slot_bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (err == hipSuccess) err = hipEventRecord(b.ev, (hipStream_t)stream);
    if (err != hipSuccess) {
      lk.lock();
      b.state = FREE; b.id = -1;
      return pn::fail(std::string("pn_spill_put: ") + hipGetErrorString(err));
    }
  } else {
    std::memcpy(b.host, src, (size_t)sp->slot_bytes);
  }
  lk.lock();
  b.state = WRITING;
  sp->jobs.push_back({0, i, id});
  lk.unlock();
  sp->cv_job.notify_one();
  return 0;
}

// mu held: the buffer that holds (or is receiving) checkpoint `id` for reading, or -1
static int find_read(pn_spill *sp, int64_t id) {
  for (size_t i = 0; i < sp->bufs.size(); ++i)
    if (sp->bufs[i].id == id && (sp->bufs[i].state == READING || sp->bufs[i].state == READY)) return (int)i;
  return -1;
}

// mu held: wait until no write of checkpoint `id` is in flight (its file is complete)
void wait_written(pn_spill *sp, std::unique_lock<std::mutex> &lk, int64_t id) {
  for (;;) {
    bool busy = false;
    for (Buf &b : sp->bufs) busy = busy || (b.id == id && (b.state == D2H_PENDING || b.state == WRITING));
    if (!busy) return;
    sp->cv_buf.wait(lk);
  }
}

// mu held: wait until no file read of checkpoint `id` is in flight
void wait_read(pn_spill *sp, std::unique_lock<std::mutex> &lk, int64_t id) {
  for (;;) {
    bool busy = false;
    for (Buf &b : sp->bufs) busy = busy || (b.id == id && b.state == READING);
    if (!busy) return;
    sp->cv_buf.wait(lk);
  }
}

int pn_spill_prefetch(pn_spill *sp, int64_t id) {
  std::unique_lock<std::mutex> lk(sp->mu);
  if (check_io(sp)) return 1;
  if (find_read(sp, id) >= 0) return 0;
  wait_written(sp, lk, id);
  if (!sp->files.count(id)) return 0;              // never written (nothing to prefetch)
  // a prefetch must not wait for a buffer: skip it when none is free
  int i = -1;
  for (size_t k = 0; k < sp->bufs.size(); ++k) {
    Buf &b = sp->bufs[k];
    if (b.state == H2D_PENDING && (!sp->device || hipEventQuery(b.ev) == hipSuccess)) { b.state = FREE; b.id = -1; }
    if (b.state == FREE) { i = (int)k; break; }
  }
  if (i < 0) return 0;
  sp->bufs[i].state = READING;
  sp->bufs[i].id = id;
  sp->jobs.push_back({1, i, id});
  lk.unlock();
  sp->cv_job.notify_one();
  return 0;
}

int pn_spill_get(pn_spill *sp, void *stream, int64_t id, void *dst) {
  std::unique_lock<std::mutex> lk(sp->mu);
  if (check_io(sp)) return 1;
  int i = find_read(sp, id);
  if (i < 0) {
    wait_written(sp, lk, id);
    if (check_io(sp)) return 1;
    if (!sp->files.count(id)) return pn::fail("pn_spill_get: checkpoint was never written");
    i = sp->take_free(lk, &sp->get_waits);
    sp->bufs[i].state = READING;
    sp->bufs[i].id = id;
    sp->jobs.push_back({1, i, id});
    sp->cv_job.notify_one();
    ++sp->get_waits;
  }
  Buf &b = sp->bufs[i];
  while (b.state == READING) sp->cv_buf.wait(lk);
  if (check_io(sp)) { b.state = FREE; b.id = -1; return 1; }
  if (sp->device) {
    b.state = H2D_PENDING;
    lk.unlock();
    hipError_t err = hipMemcpyAsync(dst, b.host, (size_t)sp->slot_bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (err == hipSuccess) err = hipEventRecord(b.ev, (hipStream_t)stream);
    if (err != hipSuccess) return pn::fail(std::string("pn_spill_get: ") + hipGetErrorString(err));
  } else {
    std::memcpy(dst, b.host, (size_t)sp->slot_bytes);
    b.state = FREE;
    b.id = -1;
  }
  return 0;
}

int pn_spill_drop(pn_spill *sp, int64_t id) {
  std::unique_lock<std::mutex> lk(sp->mu);
  wait_written(sp, lk, id);
  wait_read(sp, lk, id);             // the file must not disappear under a read-ahead
  for (Buf &b : sp->bufs)
    if (b.id == id && b.state == READY) { b.state = FREE; b.id = -1; }
  if (sp->files.erase(id) && !sp->keep_files) std::remove(sp->path(id).c_str());
  return 0;
}

int pn_spill_stats(pn_spill *sp, int64_t *files, int64_t *bytes_written, int64_t *bytes_read, int64_t *waits) {
  std::lock_guard<std::mutex> lk(sp->mu);
  if (files) *files = (int64_t)sp->files.size();
  if (bytes_written) *bytes_written = sp->bytes_written;
  if (bytes_read) *bytes_read = sp->bytes_read;
  if (waits) *waits = sp->put_waits + sp->get_waits;
  return 0;
}

}  // extern "C"
