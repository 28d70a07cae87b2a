// certfhe/Timer.h -- stopwatch kept for umbrella-header parity with
// /root/reference/src/Timer.h.  Device work is asynchronous: stop() synchronises the
// certFHE stream first so a Timer around a GPU operation measures the operation.
#ifndef CERTFHE_TIMER_H
#define CERTFHE_TIMER_H

#include "utils.h"

using namespace std;

namespace certFHE {

class Timer {
    string name;
    std::chrono::high_resolution_clock::time_point t_start;
    double elapsed_ms;
    bool running;

  public:
    Timer(string name = "Default timer");
    virtual ~Timer();

    void start();
    double stop();          // milliseconds
    void reset();
    double stopAndPrint();
    void print();
    double getValue();
};

} // namespace certFHE

#endif
