// Timer.cpp -- stopwatch; output format of /root/reference/src/Timer.cpp:31-33.
#include "Timer.h"

#include "runtime.h"

namespace certFHE {

Timer::Timer(string pname) : name(pname), elapsed_ms(0.0), running(false) {}

Timer::~Timer() {}

void Timer::start()
{
    t_start = std::chrono::high_resolution_clock::now();
    running = true;
}

double Timer::stop()
{
    detail::syncDevice();   // GPU work is asynchronous; account for it
    std::chrono::duration<double> d = std::chrono::high_resolution_clock::now() - t_start;
    elapsed_ms = d.count() * 1000.0;
    running = false;
    return elapsed_ms;
}

void Timer::reset()
{
    t_start = std::chrono::high_resolution_clock::now();
    elapsed_ms = 0.0;
}

void Timer::print()
{
    cout << name << " : " << elapsed_ms << " ms " << endl;
    fflush(stdout);
}

double Timer::stopAndPrint()
{
    stop();
    print();
    return elapsed_ms;
}

double Timer::getValue() { return elapsed_ms; }

} // namespace certFHE
