// Online update source — the consumer side of the reference's real-time update path
// (ps.json "update_source", parsed at /root/reference/hps_backend/src/backend.cpp:262-308; behaviour
// docs/hierarchical_parameter_server.md:575-646: messages of (key, embedding vector) pairs per model table are polled,
// dispatched to the database layers in chunks of at most max_batch_size keys, retried after failure_backoff_ms when a layer
// refuses them, and committed to the source after at most max_commit_interval messages).
//
// The reference's only transport is an Apache Kafka consumer (librdkafka, not in this image).  This build keeps the consumer
// loop and puts the wire behind UpdateTransport; the transport it ships is a file tail ("type": "file_tail", "brokers" = path
// of an append-only message file): a producer appends framed messages, the consumer follows the file and remembers how far
// it has committed in <path>.offset (a restarted server resumes there, as a Kafka consumer group would).
// "kafka_message_queue" is refused at start-up (parameter_server.cpp) rather than ignored.
//
// Message frame (little-endian), MessageHeader followed by the payload:
//   model name bytes | count x int64 keys | count x dim fp32 rows
#pragma once
#include <atomic>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../common/config.h"
#include "../common/status.h"

namespace hps {

constexpr uint32_t kUpdateMagic = 0x55535048u;   // "HPSU"

struct UpdateMessageHeader {
  uint32_t magic;
  uint16_t model_len;
  uint16_t table;
  uint32_t dim;
  uint32_t count;
};
static_assert(sizeof(UpdateMessageHeader) == 16, "frame header is 16 bytes");

struct UpdateMessage {
  std::string model;
  uint32_t table = 0, dim = 0;
  std::vector<int64_t> keys;
  std::vector<float> rows;   // keys.size() x dim
};

// Serialises one message (what a producer appends to the file).
std::string EncodeUpdateMessage(const std::string& model, uint32_t table, uint32_t dim, const int64_t* keys, const float* rows, size_t n);

class UpdateTransport {
 public:
  virtual ~UpdateTransport() = default;
  // Up to `max_messages` complete messages that arrived since the last Poll, waiting at most timeout_ms for the first.
  virtual Status Poll(size_t timeout_ms, size_t max_messages, std::vector<UpdateMessage>* out) = 0;
  // The first `messages` of the messages handed out by Poll since the last Commit have been dealt with (applied, or dropped
  // with a log line): a restarted consumer must not see THEM again — and must see every later one, handed out or not.
  virtual Status Commit(size_t messages) = 0;
  // well-formed frames the transport skipped since the last call because they exceed its message bound
  virtual uint64_t TakeSkipped() { return 0; }
  // the source can never be read again (a frame that is not a frame): Poll keeps returning the same error
  virtual bool dead() const { return false; }
  virtual const char* name() const = 0;
};

// receive_buffer_size bounds one message (the reference sizes its Kafka receive buffer with it)
Status MakeFileTailTransport(const std::string& path, size_t receive_buffer_size, std::unique_ptr<UpdateTransport>* out);

// what ApplyFn returns (with Code::kOk) for a message no update filter selected: dealt with, nothing applied, nothing counted
constexpr const char* kUpdateFiltered = "filtered";

struct UpdateSourceStats {
  uint64_t messages = 0, keys = 0, dispatches = 0, commits = 0, dispatch_failures = 0, rejected_messages = 0;
};

class UpdateConsumer {
 public:
  // apply(model, table, dim, keys, rows, n): hand one chunk to the database layers (HierParameterServer::upsert_table).
  // committed(models): the models whose tables changed since the last commit (their GPU caches get the new rows).
  using ApplyFn = std::function<Status(const std::string&, uint32_t, uint32_t, const int64_t*, const float*, size_t)>;
  using CommitFn = std::function<void(const std::set<std::string>&)>;
  UpdateConsumer(const UpdateSourceParams& p, std::unique_ptr<UpdateTransport> transport, ApplyFn apply, CommitFn committed);
  ~UpdateConsumer();   // stops the thread (pending messages stay uncommitted: they are replayed after a restart)
  void Stop();         // the same, explicitly (idempotent); stats() keeps working afterwards
  UpdateSourceStats stats() const;
  // blocks until every message that was in the source when the call started has been applied and committed (tests, tools)
  Status Drain(size_t timeout_ms);

 private:
  void Run();
  UpdateSourceParams p_;
  std::unique_ptr<UpdateTransport> transport_;
  ApplyFn apply_;
  CommitFn committed_;
  std::thread thread_;
  std::mutex join_mu_;
  std::atomic<bool> stop_{false};
  mutable std::mutex mu_;
  UpdateSourceStats stats_;
  uint64_t idle_polls_ = 0;   // polls that found nothing with nothing pending (Drain waits for one to pass)
  std::string dead_;          // not empty: the source is unreadable for good (Drain reports it)
};

}  // namespace hps
