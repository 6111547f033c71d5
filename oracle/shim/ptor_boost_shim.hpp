// TEST INFRASTRUCTURE ONLY (oracle build). Minimal boost -> std mapping so that the
// reference translation units compile unmodified in a container without Boost.
// Covers exactly what /root/reference/main.cpp:7-13,444,642-654 and
// /root/reference/next_event.cpp:6-13,564,701-708 use: boost::mutex,
// boost::mutex::scoped_lock, boost::bind(&worker, int, n, Scene*, V3*), boost::thread(ctor, join).
// A launch hook lets the harness borrow the Scene that the reference's main() builds on its stack.
#pragma once
#include <mutex>
#include <thread>

namespace ptor {
struct Launch {
    int worker;
    int iterations;
    void* scene;
    void* buffer;
};
// Return true when the launch was consumed by the harness (the worker is then NOT run).
typedef bool (*thread_hook_fn)(const Launch&);
extern thread_hook_fn thread_hook;
}  // namespace ptor

namespace boost {

class mutex : public std::mutex {
public:
    typedef std::unique_lock<std::mutex> scoped_lock;
};

template <class S, class B>
struct ptor_bound4 {
    void (*fn)(int, int, S*, B*);
    int a;
    int b;
    S* s;
    B* buf;
    void operator()() const { fn(a, b, s, buf); }
};

template <class S, class B, class T>
ptor_bound4<S, B> bind(void (*fn)(int, int, S*, B*), int a, T b, S* s, B* buf) {
    ptor_bound4<S, B> r = {fn, a, (int)b, s, buf};
    return r;
}

class thread {
    std::thread t_;

public:
    template <class S, class B>
    explicit thread(ptor_bound4<S, B> b) {
        ptor::Launch l = {b.a, b.b, (void*)b.s, (void*)b.buf};
        if (ptor::thread_hook && ptor::thread_hook(l)) return;
        t_ = std::thread(b);
    }
    void join() {
        if (t_.joinable()) t_.join();
    }
    ~thread() {
        if (t_.joinable()) t_.join();
    }
};

}  // namespace boost
