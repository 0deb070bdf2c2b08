// Periodic / one-shot timer used for embedding-cache refresh.
// Role of the reference's Timer (/root/reference/hps_backend/include/timer.hpp:71-99): `start` runs a task
// every `interval` seconds on its own thread, `startonce` runs a task once after a delay.  Unlike the
// reference (detached thread polled every whole second; stop() spins until it notices) this one sleeps on
// a condition variable, so stop() returns promptly and fractional intervals work.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace hps { namespace triton {

class Timer {
 public:
  Timer() = default;
  ~Timer() { stop(); }
  Timer(const Timer&) = delete;
  Timer& operator=(const Timer&) = delete;

  void start(double interval_seconds, std::function<void()> task);        // timer.hpp:71-90
  void startonce(double delay_seconds, std::function<void()> task);       // timer.hpp:92-99
  void stop();                                                            // joins every thread it started

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  bool stopping_ = false;
  std::vector<std::thread> threads_;
};

}}  // namespace hps::triton
