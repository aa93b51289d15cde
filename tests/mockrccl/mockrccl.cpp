// tests/mockrccl/mockrccl.cpp -- TEST INFRASTRUCTURE, not part of the product: a stand-in for librccl.so.1 that carries a collective
// between PROCESSES THAT SHARE ONE GPU.  No node of the build pool has two GPUs, so the library's own collective code
// (nrtsearch_amd/csrc/dist.cpp: grouped all-gather / send-recv exchange of per-shard top-k lists, then TopDocs.merge) had only ever
// run at world = 1.  With this library on the loader's path dist.cpp's dlopen("librccl.so.1") binds to it, and two processes on one
// GPU -- each a rank with its own context and its own docid shard -- run nrtgpu_dist_search_bm25_batch / _knn_exact / _hybrid_batch
// end to end with real kernels and real data (tests/test_dist_two_ranks_gpu.py).  What it is NOT: RCCL, xGMI, or a performance
// statement -- a message is a file under /dev/shm, copied out of and into device memory with hipMemcpy.
//   ncclSend : synchronises the stream, copies the buffer to the host, writes <dir>/m_<src>_<dst>_<seq> (tmp + rename); never blocks
//   ncclRecv : waits for the peer's message with its own sequence number, CHECKS ITS SIZE against what the caller expects
//              (a protocol mismatch between the ranks is an error, not a hang), copies it into device memory
//   ncclAllGather : a send to every peer + a recv from every peer + a device copy of the rank's own block (on the caller's stream)
//   groups   : calls run at once (sends never block, so any order of sends and recvs inside a group completes)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <string>
#include <vector>

extern "C" {
struct MockId { char internal[128]; };
struct MockComm {
  int world, rank;
  std::string dir;
  std::vector<uint64_t> sent, received;   // per peer: messages so far
};
typedef MockComm* ncclComm_t;

static size_t dtype_bytes(int dt) {   // rccl.h: ncclDataType_t
  switch (dt) {
    case 0: case 1: return 1;              // int8 / uint8
    case 2: case 3: case 7: return 4;      // int32 / uint32 / float32
    case 4: case 5: case 8: return 8;      // int64 / uint64 / float64
    case 6: case 9: return 2;              // float16 / bfloat16
    default: return 0;
  }
}
static double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static bool wait_for_file(const std::string& path, double timeout_s) {
  const double t0 = now_s();
  struct stat st;
  while (stat(path.c_str(), &st) != 0) {
    if (now_s() - t0 > timeout_s) return false;
    usleep(200);
  }
  return true;
}

int ncclGetUniqueId(MockId* id) {
  memset(id, 0, sizeof(*id));
  uint64_t r[2] = {(uint64_t)getpid() * 0x9E3779B97F4A7C15ull ^ (uint64_t)(now_s() * 1e9), 0};
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd >= 0) {
    (void)!read(fd, r, sizeof(r));
    close(fd);
  }
  snprintf(id->internal, sizeof(id->internal), "%016llx%016llx", (unsigned long long)r[0], (unsigned long long)r[1]);
  return 0;
}

int ncclCommInitRank(ncclComm_t* comm, int world, MockId id, int rank) {
  if (!comm || world <= 0 || rank < 0 || rank >= world) return 4;   // ncclInvalidArgument
  MockComm* c = new MockComm();
  c->world = world;
  c->rank = rank;
  c->dir = std::string("/dev/shm/nrtgpu_mockrccl_") + std::string(id.internal, strnlen(id.internal, 32));
  c->sent.assign((size_t)world, 0);
  c->received.assign((size_t)world, 0);
  (void)mkdir(c->dir.c_str(), 0700);
  {
    const std::string mine = c->dir + "/joined_" + std::to_string(rank);
    int fd = open(mine.c_str(), O_CREAT | O_WRONLY, 0600);
    if (fd < 0) { delete c; return 2; }   // ncclSystemError
    close(fd);
  }
  for (int r = 0; r < world; ++r)
    if (!wait_for_file(c->dir + "/joined_" + std::to_string(r), 180.0)) { delete c; return 2; }
  *comm = c;
  return 0;
}

int ncclCommDestroy(ncclComm_t c) {
  if (!c) return 0;
  const std::string left = c->dir + "/left_" + std::to_string(c->rank);
  int fd = open(left.c_str(), O_CREAT | O_WRONLY, 0600);
  if (fd >= 0) close(fd);
  if (c->rank == 0) {   // the last one out would be better; rank 0 waits a little for the others and removes what is there
    for (int r = 1; r < c->world; ++r) (void)wait_for_file(c->dir + "/left_" + std::to_string(r), 10.0);
    for (int r = 0; r < c->world; ++r) {
      (void)unlink((c->dir + "/joined_" + std::to_string(r)).c_str());
      (void)unlink((c->dir + "/left_" + std::to_string(r)).c_str());
    }
    (void)rmdir(c->dir.c_str());
  }
  delete c;
  return 0;
}

int ncclSend(const void* buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t stream) {
  const size_t bytes = count * dtype_bytes(dtype);
  if (!c || peer < 0 || peer >= c->world || dtype_bytes(dtype) == 0) return 4;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;   // ncclUnhandledCudaError
  std::vector<char> host(bytes);
  if (bytes && hipMemcpy(host.data(), buf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  const std::string name = c->dir + "/m_" + std::to_string(c->rank) + "_" + std::to_string(peer) + "_" + std::to_string(c->sent[(size_t)peer]++);
  const std::string tmp = name + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return 2;
  const bool ok = bytes == 0 || fwrite(host.data(), 1, bytes, f) == bytes;
  fclose(f);
  if (!ok || rename(tmp.c_str(), name.c_str()) != 0) return 2;
  return 0;
}

int ncclRecv(void* buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t stream) {
  const size_t bytes = count * dtype_bytes(dtype);
  if (!c || peer < 0 || peer >= c->world || dtype_bytes(dtype) == 0) return 4;
  const std::string name = c->dir + "/m_" + std::to_string(peer) + "_" + std::to_string(c->rank) + "_" + std::to_string(c->received[(size_t)peer]++);
  if (!wait_for_file(name, 180.0)) return 6;   // ncclRemoteError: the peer never sent its part
  struct stat st;
  if (stat(name.c_str(), &st) != 0) return 2;
  if ((size_t)st.st_size != bytes) {
    fprintf(stderr, "[mockrccl] rank %d expects %zu bytes from rank %d, the peer sent %zu: the ranks disagree about the exchange\n", c->rank, bytes, peer,
            (size_t)st.st_size);
    return 5;   // ncclInvalidUsage
  }
  std::vector<char> host(bytes);
  FILE* f = fopen(name.c_str(), "rb");
  if (!f) return 2;
  const bool ok = bytes == 0 || fread(host.data(), 1, bytes, f) == bytes;
  fclose(f);
  (void)unlink(name.c_str());
  if (!ok) return 2;
  // ON THE CALLER'S STREAM, like the collective this stands in for: the library waits for that stream and then reads the buffer
  // from another one (see ncclAllGather below)
  if (bytes && hipMemcpyAsync(buf, host.data(), bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;   // (`host` goes out of scope)
  return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, ncclComm_t c, hipStream_t stream) {
  const size_t bytes = count * dtype_bytes(dtype);
  if (!c || dtype_bytes(dtype) == 0) return 4;
  for (int p = 0; p < c->world; ++p)
    if (p != c->rank)
      if (int rc = ncclSend(send, count, dtype, p, c, stream)) return rc;
  for (int p = 0; p < c->world; ++p) {
    char* dst = (char*)recv + (size_t)p * bytes;
    if (p == c->rank) {
      // The rank's own block: copied ON THE CALLER'S STREAM.  Through round 5 this was a plain hipMemcpy(..., DeviceToDevice): a
      // device-to-device hipMemcpy returns before the copy has run and is ordered against the NULL stream only -- dist.cpp waits
      // for ITS stream (RCCL's contract) and then launches TopDocs.merge on another non-blocking stream, which could read the
      // PREVIOUS exchange's own block.  The two ranks then merged different lists, reached different verdicts on the shards'
      // speculative thresholds, and one of them entered the re-run's collective alone: the "unexplained failure of the N = 2 bench
      // loop, all-gather form only" of round 5 (3 of 32 runs of scripts/gpu_two_rank_loop.sh before this fix; the all-to-all form
      // copies its own slice itself, on its stream, and never failed).  A race of this stand-in, not of the library.
      if (bytes && hipMemcpyAsync(dst, send, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
      if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    } else if (int rc = ncclRecv(dst, count, dtype, p, c, stream)) {
      return rc;
    }
  }
  return 0;
}

int ncclGroupStart(void) { return 0; }
int ncclGroupEnd(void) { return 0; }
const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case 0: return "success";
    case 1: return "mockrccl: HIP error";
    case 2: return "mockrccl: system error (a /dev/shm file)";
    case 4: return "mockrccl: invalid argument";
    case 5: return "mockrccl: the ranks disagree about a message's size";
    case 6: return "mockrccl: the peer never sent its part (timeout)";
    default: return "mockrccl: error";
  }
}
}
