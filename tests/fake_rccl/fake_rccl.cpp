// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl on boxes with ONE GPU (tests/test_gpu_multirank_standin.py).
//
// RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so on a one-GPU test box the product's multi-rank C++ path
// (prisim_amd/csrc/capi.cpp: communicator setup, gather_one_slot, the gathered-cube layout, complex64 send buffers, lag / gradient
// gathers, gather-to-root, the self-test, the event timing) could never run with more than one rank.  libprisim_hip.so dlopen()s its
// RCCL, and PRISIM_RCCL_LIB names another library to load instead: this one.  It implements the seven entry points the product uses
// with the real signatures of <rccl/rccl.h>, moving DEVICE buffers between processes through files in a directory named by the
// "unique id": every call first waits for the work queued on its stream, copies device -> host -> file, waits for the peers' files and
// copies host -> device.  Synchronous and slow, but byte-exact, and every rank runs the real product code around it.
// It never ships: nothing under prisim_amd/ refers to it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>
#include <unistd.h>

struct ncclComm {
  std::string dir;
  int nranks = 1, rank = 0;
  long seq = 0;
  struct P2P { bool send; const void* sbuf; void* rbuf; size_t bytes; int peer; hipStream_t stream; };
  std::vector<P2P> group;
  int group_depth = 0;
};

namespace {

thread_local ncclComm* g_group_comm = nullptr;
int g_group_depth = 0;

size_t dtype_size(ncclDataType_t t) {
  switch ((int)t) {
    case 0: case 1: return 1;
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
    case 6: case 9: return 2;
    default: return 0;
  }
}

bool write_file(const std::string& path, const void* data, size_t bytes) {
  const std::string tmp = path + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = bytes == 0 || fwrite(data, 1, bytes, f) == bytes;
  fclose(f);
  return ok && rename(tmp.c_str(), path.c_str()) == 0;
}

bool read_file_when_there(const std::string& path, void* data, size_t bytes, double timeout_s = 300.0) {
  const auto t0 = std::chrono::steady_clock::now();
  struct stat st;
  while (stat(path.c_str(), &st) != 0) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  const bool ok = bytes == 0 || fread(data, 1, bytes, f) == bytes;
  fclose(f);
  return ok;
}

std::string op_name(const ncclComm* c, const char* kind, long seq, int a, int b = -1) {
  char buf[256];
  snprintf(buf, sizeof(buf), "%s/%s_%ld_%d_%d.bin", c->dir.c_str(), kind, seq, a, b);
  return buf;
}

ncclResult_t run_p2p(ncclComm* c) {
  const long seq = c->seq++;
  std::vector<char> host;
  for (const auto& op : c->group) {                 // sends first: nothing here blocks on a peer
    if (!op.send) continue;
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    host.resize(op.bytes);
    if (hipMemcpy(host.data(), op.sbuf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!write_file(op_name(c, "p2p", seq, c->rank, op.peer), host.data(), op.bytes)) return ncclSystemError;
  }
  for (const auto& op : c->group) {
    if (op.send) continue;
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    host.resize(op.bytes);
    if (!read_file_when_there(op_name(c, "p2p", seq, op.peer, c->rank), host.data(), op.bytes)) return ncclSystemError;
    if (hipMemcpy(op.rbuf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  }
  c->group.clear();
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof(*id));
  char tmpl[] = "/tmp/prisim_fake_rccl_XXXXXX";
  if (!mkdtemp(tmpl)) return ncclSystemError;
  strncpy(id->internal, tmpl, sizeof(id->internal) - 1);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->dir = std::string(id.internal, strnlen(id.internal, sizeof(id.internal)));
  c->nranks = nranks; c->rank = rank;
  char dummy = 1;
  if (!write_file(op_name(c, "init", 0, rank), &dummy, 1)) { delete c; return ncclSystemError; }
  for (int r = 0; r < nranks; ++r)
    if (!read_file_when_there(op_name(c, "init", 0, r), &dummy, 1)) { delete c; return ncclSystemError; }
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete comm;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "stand-in RCCL error"; }

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t c, hipStream_t stream) {
  if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
  const size_t bytes = sendcount * dtype_size(datatype);
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<char> host(bytes);
  if (hipMemcpy(host.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  const long seq = c->seq++;
  if (!write_file(op_name(c, "ag", seq, c->rank), host.data(), bytes)) return ncclSystemError;
  for (int r = 0; r < c->nranks; ++r) {
    if (!read_file_when_there(op_name(c, "ag", seq, r), host.data(), bytes)) return ncclSystemError;
    if (hipMemcpy((char*)recvbuff + (size_t)r * bytes, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_group_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
  if (g_group_depth <= 0) return ncclInvalidUsage;
  if (--g_group_depth > 0) return ncclSuccess;
  ncclComm* c = g_group_comm;
  g_group_comm = nullptr;
  return c ? run_p2p(c) : ncclSuccess;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t c, hipStream_t stream) {
  if (!c || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
  c->group.push_back({true, sendbuff, nullptr, count * dtype_size(datatype), peer, stream});
  if (g_group_depth > 0) { g_group_comm = c; return ncclSuccess; }
  return run_p2p(c);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t c, hipStream_t stream) {
  if (!c || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
  c->group.push_back({false, nullptr, recvbuff, count * dtype_size(datatype), peer, stream});
  if (g_group_depth > 0) { g_group_comm = c; return ncclSuccess; }
  return run_p2p(c);
}

}  // extern "C"
