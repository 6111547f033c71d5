// TEST INFRASTRUCTURE ONLY.  Minimal boost -> std mapping: boost::mutex, boost::mutex::scoped_lock,
// boost::bind(&worker, int, n, Scene*, V3*), boost::thread(ctor, join), i.e. what
// /root/reference/main.cpp:7-13,444,642-654 and next_event.cpp:6-13,564,701-708 use.
// NOT part of the parity oracle: the reference builds that produce golden vectors and check the
// restatement (oracle/_ref/libptref*.so) contain no Boost and no stand-in for it (oracle/Makefile).
// Two things still read the reference with its thread glue and therefore go through this header:
// the literal multi-threaded next_event run that is only ever timed (ref_literal.cpp), and the
// syntax check of the binding stub a maintainer with Boost would add (binding_stub_check.cpp).
#pragma once
#include <mutex>
#include <thread>

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
    explicit thread(ptor_bound4<S, B> b) : t_(b) {}
    void join() {
        if (t_.joinable()) t_.join();
    }
    ~thread() {
        if (t_.joinable()) t_.join();
    }
};

}  // namespace boost
