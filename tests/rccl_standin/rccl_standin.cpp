// TEST-ONLY stand-in for librccl.so: the handful of entry points the slab drivers
// use (soda_hip.cpp: soda_hip_run_slab; the generated <app>_multi_gpu), implemented
// for "ranks" that are host threads of ONE process sharing ONE GPU.  RCCL itself
// refuses two ranks on a device, and no multi-GPU box is available to the GPU
// tests, so without this the world > 1 code below the C ABI would never execute.
//
// A send and its matching receive become one stream-ordered device-to-device copy:
//   sender   : records `ready` on its stream (the rows exist), posts the message;
//   receiver : waits (host) for the post, makes its stream wait for `ready`,
//              enqueues the copy, records `done`;
//   sender   : waits (host) for that, makes its stream wait for `done` (the source
//              rows may be overwritten only after the copy).
// ncclGroupStart/End batch operations exactly as the callers use them: all sends of
// a group are posted before any receive blocks, so an open chain of any length
// cannot deadlock.  ncclCommAbort is LOCAL, as RCCL's is: it fails the pending and later
// operations of the communicator it is given and of no other - a caller that wants its
// peers unblocked aborts THEIR communicators too (which a one-process driver can).  An
// aborted communicator's memory is not released here (another thread may be inside an
// operation on it; RCCL handles that internally, a test library can simply leak).
// Built by tests/test_gpu_parity.py with -Wl,-soname,librccl.so and
// loaded before libsoda_hip resolves "librccl.so" (the loader then hands this object
// back by its soname); it is never on a product path.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace {

struct Message {
  const void* src = nullptr;
  size_t bytes = 0;
  hipEvent_t ready = nullptr, done = nullptr;
  bool copied = false, failed = false;
};

struct World {
  std::mutex m;
  std::condition_variable cv;
  int n = 0, alive = 0;
  bool aborted = false;
  std::map<std::pair<int, int>, std::deque<Message*>> box;   // (from, to)
  long long messages = 0, bytes = 0;
};

struct Comm {
  World* world;
  int rank;
  bool aborted = false;     // under world->m
};

struct Op {
  bool send;
  void* buf;
  size_t bytes;
  int peer;
  Comm* comm;
  hipStream_t stream;
};

thread_local std::vector<Op> t_ops;
thread_local int t_depth = 0;

enum { kSuccess = 0, kUnhandledCudaError = 1, kSystemError = 2, kInvalidArgument = 4 };

int flush() {
  std::vector<Op> ops;
  ops.swap(t_ops);
  std::vector<std::pair<Message*, Op>> mine;
  int rc = kSuccess;
  for (const Op& op : ops) {   // 1. post every send
    if (!op.send) continue;
    Message* msg = new Message;
    msg->src = op.buf;
    msg->bytes = op.bytes;
    if (hipEventCreateWithFlags(&msg->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(msg->ready, op.stream) != hipSuccess)
      return kUnhandledCudaError;
    World* w = op.comm->world;
    {
      std::lock_guard<std::mutex> lock(w->m);
      w->box[{op.comm->rank, op.peer}].push_back(msg);
      w->messages += 1;
      w->bytes += (long long)op.bytes;
    }
    w->cv.notify_all();
    mine.push_back({msg, op});
  }
  for (const Op& op : ops) {   // 2. receive: one copy on the receiver's stream each
    if (op.send) continue;
    World* w = op.comm->world;
    Message* msg = nullptr;
    {
      std::unique_lock<std::mutex> lock(w->m);
      auto& queue = w->box[{op.peer, op.comm->rank}];
      w->cv.wait(lock, [&] { return op.comm->aborted || !queue.empty(); });
      if (op.comm->aborted || queue.empty()) { rc = kSystemError; continue; }
      msg = queue.front();
      queue.pop_front();
    }
    bool ok = msg->bytes == op.bytes;
    if (!ok) fprintf(stderr, "rccl stand-in: rank %d expects %zu bytes from %d, it sends %zu\n",
                     op.comm->rank, op.bytes, op.peer, msg->bytes);
    ok = ok && hipStreamWaitEvent(op.stream, msg->ready, 0) == hipSuccess &&
         hipMemcpyAsync(op.buf, msg->src, op.bytes, hipMemcpyDeviceToDevice, op.stream) == hipSuccess &&
         hipEventCreateWithFlags(&msg->done, hipEventDisableTiming) == hipSuccess &&
         hipEventRecord(msg->done, op.stream) == hipSuccess;
    {
      std::lock_guard<std::mutex> lock(w->m);
      msg->copied = true;
      msg->failed = !ok;
    }
    w->cv.notify_all();
    if (!ok) rc = kUnhandledCudaError;
  }
  for (auto& entry : mine) {   // 3. the source rows are free once the copy has run
    Message* msg = entry.first;
    World* w = entry.second.comm->world;
    {
      std::unique_lock<std::mutex> lock(w->m);
      w->cv.wait(lock, [&] { return entry.second.comm->aborted || msg->copied; });
      if (!msg->copied) { rc = kSystemError; continue; }   // (message stays queued: leaked)
    }
    if (msg->failed || hipStreamWaitEvent(entry.second.stream, msg->done, 0) != hipSuccess)
      rc = kUnhandledCudaError;
    (void)hipEventDestroy(msg->ready);
    if (msg->done) (void)hipEventDestroy(msg->done);
    delete msg;
  }
  return rc;
}

int enqueue(bool send, void* buf, size_t count, int datatype, int peer, void* comm,
            hipStream_t stream) {
  static const size_t width[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};   // ncclInt8 .. ncclBfloat16
  if (!comm || !buf || datatype < 0 || datatype > 9) return kInvalidArgument;
  Comm* c = (Comm*)comm;
  if (peer < 0 || peer >= c->world->n || peer == c->rank) return kInvalidArgument;
  {
    std::lock_guard<std::mutex> lock(c->world->m);
    if (c->aborted) return kSystemError;
  }
  t_ops.push_back(Op{send, buf, count * width[datatype], peer, c, stream});
  return t_depth > 0 ? kSuccess : flush();
}

}  // namespace

extern "C" {

int ncclGroupStart() { ++t_depth; return kSuccess; }
int ncclGroupEnd() {
  if (t_depth <= 0) return kInvalidArgument;
  return --t_depth == 0 ? flush() : kSuccess;
}
int ncclSend(const void* buf, size_t count, int datatype, int peer, void* comm,
             hipStream_t stream) {
  return enqueue(true, const_cast<void*>(buf), count, datatype, peer, comm, stream);
}
int ncclRecv(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream) {
  return enqueue(false, buf, count, datatype, peer, comm, stream);
}
const char* ncclGetErrorString(int code) {
  switch (code) {
    case kSuccess: return "no error";
    case kUnhandledCudaError: return "unhandled hip error (rccl stand-in)";
    case kSystemError: return "communicator aborted (rccl stand-in)";
    case kInvalidArgument: return "invalid argument (rccl stand-in)";
    default: return "unknown (rccl stand-in)";
  }
}
int ncclCommInitAll(void** comms, int n, const int* /*devices*/) {
  if (!comms || n < 1) return kInvalidArgument;
  World* w = new World;
  w->n = w->alive = n;
  for (int r = 0; r < n; ++r) comms[r] = new Comm{w, r, false};
  return kSuccess;
}
static int release(void* comm, bool abort) {
  if (!comm) return kInvalidArgument;
  Comm* c = (Comm*)comm;
  World* w = c->world;
  if (abort) {
    {
      std::lock_guard<std::mutex> lock(w->m);
      if (c->aborted) {
        fprintf(stderr, "rccl stand-in: communicator of rank %d aborted twice\n", c->rank);
        w->aborted = true;      // (kept as a flag tests can read: a double abort happened)
        return kInvalidArgument;
      }
      c->aborted = true;
    }
    w->cv.notify_all();
    return kSuccess;            // nothing freed: see the header comment
  }
  bool last;
  {
    std::lock_guard<std::mutex> lock(w->m);
    last = --w->alive == 0;
  }
  w->cv.notify_all();
  delete c;
  if (last) delete w;
  return kSuccess;
}
int ncclCommDestroy(void* comm) { return release(comm, false); }
int ncclCommAbort(void* comm) { return release(comm, true); }
// test instrumentation: did anybody abort a communicator a second time?
int rccl_standin_double_abort(void* comm) {
  if (!comm) return -1;
  World* w = ((Comm*)comm)->world;
  std::lock_guard<std::mutex> lock(w->m);
  return w->aborted ? 1 : 0;
}
// test instrumentation: how much went through (proves the exchange ran)
int rccl_standin_traffic(void* comm, long long* messages, long long* bytes) {
  if (!comm) return kInvalidArgument;
  World* w = ((Comm*)comm)->world;
  std::lock_guard<std::mutex> lock(w->m);
  if (messages) *messages = w->messages;
  if (bytes) *bytes = w->bytes;
  return kSuccess;
}

}  // extern "C"
