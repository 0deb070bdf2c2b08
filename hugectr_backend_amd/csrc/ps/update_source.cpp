#include "update_source.h"

#include <deque>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>

namespace hps {

std::string EncodeUpdateMessage(const std::string& model, uint32_t table, uint32_t dim, const int64_t* keys, const float* rows, size_t n) {
  UpdateMessageHeader h{kUpdateMagic, (uint16_t)model.size(), (uint16_t)table, dim, (uint32_t)n};
  std::string out;
  out.resize(sizeof h + model.size() + n * sizeof(int64_t) + n * (size_t)dim * sizeof(float));
  char* p = &out[0];
  memcpy(p, &h, sizeof h); p += sizeof h;
  memcpy(p, model.data(), model.size()); p += model.size();
  memcpy(p, keys, n * sizeof(int64_t)); p += n * sizeof(int64_t);
  memcpy(p, rows, n * (size_t)dim * sizeof(float));
  return out;
}

namespace {

// Follows an append-only file of framed messages.  A frame that is not complete yet (the producer is in the middle of an
// append) is left for the next poll; a frame that cannot be one (bad magic, no row width, an absurd length) poisons the
// source: the consumer reports it once and stops reading rather than guessing where the next frame starts.  A well-formed
// frame that is merely larger than the receive buffer is stepped over (its length is known).
class FileTailTransport : public UpdateTransport {
 public:
  FileTailTransport(std::string path, size_t max_message) : path_(std::move(path)), max_message_(std::max<size_t>(max_message, 4096)) {
    // resume behind the last commit
    if (FILE* f = fopen((path_ + ".offset").c_str(), "r")) {
      unsigned long long v = 0;
      if (fscanf(f, "%llu", &v) == 1) committed_ = read_ = (uint64_t)v;
      fclose(f);
    }
  }
  ~FileTailTransport() override { if (fd_ >= 0) close(fd_); }
  const char* name() const override { return "file_tail"; }

  Status Poll(size_t timeout_ms, size_t max_messages, std::vector<UpdateMessage>* out) override {
    out->clear();
    if (poisoned_) return Error(Code::kInternal, "update source '", path_, "': unreadable frame at offset ", read_);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    for (;;) {
      HPS_RETURN_IF_ERROR(ReadSome(max_messages, out));
      if (!out->empty() || std::chrono::steady_clock::now() >= deadline) return Status::Ok();
      usleep(2000);
    }
  }

  Status Commit(size_t messages) override {
    // only what the consumer has dealt with: the end of the `messages`-th message handed out since the last commit
    uint64_t upto = committed_;
    for (size_t i = 0; i < messages && !ends_.empty(); ++i) { upto = ends_.front(); ends_.pop_front(); }
    if (upto == committed_) return Status::Ok();
    const std::string tmp = path_ + ".offset.tmp";
    FILE* f = fopen(tmp.c_str(), "w");
    if (!f) return Error(Code::kInternal, "update source: cannot write '", tmp, "'");
    fprintf(f, "%llu\n", (unsigned long long)upto);
    fclose(f);
    if (rename(tmp.c_str(), (path_ + ".offset").c_str()) != 0) return Error(Code::kInternal, "update source: cannot replace '", path_, ".offset'");
    committed_ = upto;
    return Status::Ok();
  }
  uint64_t TakeSkipped() override { const uint64_t n = skipped_; skipped_ = 0; return n; }
  bool dead() const override { return poisoned_; }

 private:
  Status ReadSome(size_t max_messages, std::vector<UpdateMessage>* out) {
    if (fd_ < 0) {
      fd_ = open(path_.c_str(), O_RDONLY | O_CLOEXEC);
      if (fd_ < 0) return Status::Ok();   // the producer has not created it yet
    }
    struct stat st;
    if (fstat(fd_, &st) != 0) return Error(Code::kInternal, "update source: fstat('", path_, "') failed");
    uint64_t size = (uint64_t)st.st_size;
    while (out->size() < max_messages && read_ + sizeof(UpdateMessageHeader) <= size) {
      UpdateMessageHeader h;
      if (pread(fd_, &h, sizeof h, (off_t)read_) != (ssize_t)sizeof h) break;
      const uint64_t payload = (uint64_t)h.model_len + (uint64_t)h.count * sizeof(int64_t) + (uint64_t)h.count * h.dim * sizeof(float);
      if (h.magic != kUpdateMagic || h.dim == 0 || payload > (1ull << 32)) {   // (a length nobody would wait for: not a frame)
        poisoned_ = true;
        return Error(Code::kInternal, "update source '", path_, "': frame at offset ", read_, " is not a message (magic ", h.magic,
                     ", dim ", h.dim, ")");
      }
      if (read_ + sizeof h + payload > size) break;   // still being appended
      if (payload > max_message_) {
        // a well-formed message, only larger than receive_buffer_size allows: its length is known, so it is stepped over
        // (one rejection, one log line) and the source lives on
        fprintf(stderr, "[hps update source] '%s': message of %llu payload bytes at offset %llu skipped (receive_buffer_size bounds a message at %zu)\n",
                path_.c_str(), (unsigned long long)payload, (unsigned long long)read_, max_message_);
        ++skipped_;
        read_ += sizeof h + payload;
        continue;
      }
      buf_.resize((size_t)payload);
      if (payload && pread(fd_, buf_.data(), (size_t)payload, (off_t)(read_ + sizeof h)) != (ssize_t)payload) break;
      UpdateMessage m;
      m.model.assign(buf_.data(), h.model_len);
      m.table = h.table;
      m.dim = h.dim;
      m.keys.resize(h.count);
      m.rows.resize((size_t)h.count * h.dim);
      memcpy(m.keys.data(), buf_.data() + h.model_len, (size_t)h.count * sizeof(int64_t));
      memcpy(m.rows.data(), buf_.data() + h.model_len + (size_t)h.count * sizeof(int64_t), m.rows.size() * sizeof(float));
      out->push_back(std::move(m));
      read_ += sizeof h + payload;
      ends_.push_back(read_);
    }
    return Status::Ok();
  }

  std::string path_;
  size_t max_message_;
  int fd_ = -1;
  uint64_t read_ = 0, committed_ = 0;
  std::deque<uint64_t> ends_;   // end offsets of the messages handed out and not committed yet, in order
  uint64_t skipped_ = 0;
  bool poisoned_ = false;
  std::vector<char> buf_;
};

}  // namespace

Status MakeFileTailTransport(const std::string& path, size_t receive_buffer_size, std::unique_ptr<UpdateTransport>* out) {
  if (path.empty()) return Error(Code::kInvalidArg, "update_source.type = file_tail needs the message file's path in 'brokers'");
  out->reset(new FileTailTransport(path, receive_buffer_size));
  return Status::Ok();
}

UpdateConsumer::UpdateConsumer(const UpdateSourceParams& p, std::unique_ptr<UpdateTransport> transport, ApplyFn apply, CommitFn committed)
    : p_(p), transport_(std::move(transport)), apply_(std::move(apply)), committed_(std::move(committed)) {
  thread_ = std::thread([this] { Run(); });
}

UpdateConsumer::~UpdateConsumer() { Stop(); }

void UpdateConsumer::Stop() {
  stop_.store(true);
  std::lock_guard<std::mutex> lk(join_mu_);
  if (thread_.joinable()) thread_.join();
}

UpdateSourceStats UpdateConsumer::stats() const {
  std::lock_guard<std::mutex> lk(mu_);
  return stats_;
}

Status UpdateConsumer::Drain(size_t timeout_ms) {
  // two idle polls: the one in progress when the call started may have begun before the producer's last append.
  // (Polled, not waited for on the condition variable: wait_for on the steady clock is pthread_cond_clockwait, which the
  //  ThreadSanitizer runtime of this toolchain does not know — it then believes the mutex stays locked through the wait.)
  uint64_t seen;
  { std::lock_guard<std::mutex> lk(mu_); seen = idle_polls_; }
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
  for (;;) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (!dead_.empty()) return Error(Code::kInternal, dead_);
      if (idle_polls_ >= seen + 2) return Status::Ok();
    }
    if (std::chrono::steady_clock::now() >= deadline) return Error(Code::kUnavailable, "update source: still busy after ", timeout_ms, " ms");
    usleep(1000);
  }
}

void UpdateConsumer::Run() {
  std::set<std::string> touched;
  size_t since_commit = 0;   // messages dealt with (applied, or dropped with a log line) since the last commit
  auto commit = [&] {
    if (since_commit == 0) return;
    if (committed_) committed_(touched);   // the GPU caches take the new rows before the source forgets the messages
    if (transport_->Commit(since_commit).ok()) {
      std::lock_guard<std::mutex> lk(mu_);
      ++stats_.commits;
    }
    touched.clear();
    since_commit = 0;
  };
  std::vector<UpdateMessage> msgs;
  const size_t poll_ms = std::max<size_t>(1, std::min<size_t>(p_.poll_timeout_ms, 100));   // short polls: the stop flag is looked at between them
  size_t waited_ms = 0;
  while (!stop_.load()) {
    const size_t room = std::max<size_t>(1, p_.max_commit_interval) - std::min(since_commit, std::max<size_t>(1, p_.max_commit_interval) - 1);
    const Status ps = transport_->Poll(poll_ms, room, &msgs);
    if (const uint64_t sk = transport_->TakeSkipped()) { std::lock_guard<std::mutex> lk(mu_); stats_.rejected_messages += sk; }
    if (!ps.ok()) {
      if (transport_->dead()) {
        // unreadable for good: counted and logged once; what was applied before is delivered, then the thread only waits for
        // the server to stop (Drain reports the error instead of calling a dead source drained)
        bool first;
        { std::lock_guard<std::mutex> lk(mu_); first = dead_.empty(); if (first) { dead_ = ps.message(); ++stats_.rejected_messages; } }
        if (first) fprintf(stderr, "[hps update source] %s\n", ps.message().c_str());
        commit();
        usleep((useconds_t)std::max<size_t>(1, std::min<size_t>(p_.failure_backoff_ms, 100)) * 1000);
        continue;
      }
      { std::lock_guard<std::mutex> lk(mu_); ++stats_.rejected_messages; }
      fprintf(stderr, "[hps update source] %s\n", ps.message().c_str());
      commit();
      usleep((useconds_t)std::max<size_t>(1, p_.failure_backoff_ms) * 1000);
      { std::lock_guard<std::mutex> lk(mu_); ++idle_polls_; }
      continue;
    }
    if (msgs.empty()) {
      waited_ms += poll_ms;
      // nothing more came within poll_timeout_ms: what has been applied is delivered now (docs: "maximum time to wait for
      // additional updates before dispatching")
      if (waited_ms >= p_.poll_timeout_ms || since_commit == 0) {
        commit();
        waited_ms = 0;
        { std::lock_guard<std::mutex> lk(mu_); ++idle_polls_; }
      }
      continue;
    }
    waited_ms = 0;
    for (const UpdateMessage& m : msgs) {
      // Shutting down: this message and every later one of the poll stay UNCOMMITTED (they are replayed after a restart) and
      // nothing is counted for them.  What HAS been applied since the last commit is delivered like any commit: the GPU caches
      // take the new rows (the server keeps serving after the stop; round 4 left the caches with the old rows of keys the host
      // tier had already updated) and the offset moves past exactly those messages.
      if (stop_.load()) { commit(); return; }
      const size_t n = m.keys.size();
      const size_t chunk = std::max<size_t>(1, p_.max_batch_size);
      bool ok = true, cut_short = false, filtered = false;
      for (size_t b = 0; b < n && ok && !filtered; b += chunk) {
        if (stop_.load()) { cut_short = true; break; }
        const size_t e = std::min(n, b + chunk);
        // a layer that refuses the chunk is asked again after failure_backoff_ms (three times; then the message is dropped
        // with a log line: a message for a model this server does not hold would otherwise block the source for ever)
        Status st;
        for (int attempt = 0; attempt < 3; ++attempt) {
          st = apply_(m.model, m.table, m.dim, m.keys.data() + b, m.rows.data() + b * (size_t)m.dim, e - b);
          { std::lock_guard<std::mutex> lk(mu_); ++stats_.dispatches; if (!st.ok()) ++stats_.dispatch_failures; }
          if (st.ok() || st.code() == Code::kNotFound || st.code() == Code::kInvalidArg) break;
          usleep((useconds_t)std::max<size_t>(1, p_.failure_backoff_ms) * 1000);
        }
        if (st.ok() && st.message() == kUpdateFiltered) filtered = true;   // no database layer subscribed to it: nothing to count
        if (!st.ok()) {
          ok = false;
          std::lock_guard<std::mutex> lk(mu_);
          ++stats_.rejected_messages;
          fprintf(stderr, "[hps update source] message for model '%s' table %u dropped: %s\n", m.model.c_str(), m.table, st.message().c_str());
        }
      }
      if (cut_short) {
        // partly applied (upserts are idempotent): replayed whole after a restart, NOT committed — but the chunks that did reach
        // the host tier must reach the GPU caches too
        commit();
        if (committed_) committed_({m.model});
        return;
      }
      if (ok && !filtered) {
        touched.insert(m.model);
        std::lock_guard<std::mutex> lk(mu_);
        ++stats_.messages;
        stats_.keys += n;
      }
      if (++since_commit >= std::max<size_t>(1, p_.max_commit_interval)) commit();
    }
  }
  commit();   // stop seen between polls: everything applied so far is delivered and committed
}

}  // namespace hps
