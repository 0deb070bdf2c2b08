#include "timer.h"

#include <chrono>

namespace hps { namespace triton {

void Timer::start(double interval_seconds, std::function<void()> task) {
  if (!(interval_seconds > 0)) return;
  std::lock_guard<std::mutex> lk(mu_);
  stopping_ = false;
  threads_.emplace_back([this, interval_seconds, task = std::move(task)] {
    std::unique_lock<std::mutex> lk(mu_);
    const auto period = std::chrono::duration<double>(interval_seconds);
    for (;;) {
      if (cv_.wait_for(lk, period, [this] { return stopping_; })) return;
      lk.unlock();
      task();
      lk.lock();
    }
  });
}

void Timer::startonce(double delay_seconds, std::function<void()> task) {
  std::lock_guard<std::mutex> lk(mu_);
  stopping_ = false;
  threads_.emplace_back([this, delay_seconds, task = std::move(task)] {
    {
      std::unique_lock<std::mutex> lk(mu_);
      if (delay_seconds > 0 &&
          cv_.wait_for(lk, std::chrono::duration<double>(delay_seconds), [this] { return stopping_; }))
        return;
      if (stopping_) return;
    }
    task();
  });
}

void Timer::stop() {
  std::vector<std::thread> ths;
  {
    std::lock_guard<std::mutex> lk(mu_);
    stopping_ = true;
    ths.swap(threads_);
  }
  cv_.notify_all();
  for (auto& t : ths) if (t.joinable()) t.join();
}

}}  // namespace hps::triton
