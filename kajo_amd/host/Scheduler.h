// Scheduler.h -- Kajo's backend plugin interface (renderer/Scheduler.h:12-16 in the reference):
// a backend is constructed as X::Scheduler(const scene::Scene&, Image*, Preview*) and run().
// In a Kajo checkout hip::Scheduler derives from the reference's own ::Scheduler; this header is
// the stand-in that lets the backend and its headless driver build without that tree.
#ifndef KAJO_HOST_SCHEDULER_H
#define KAJO_HOST_SCHEDULER_H

class Scheduler
{
public:
    virtual ~Scheduler() {}
    virtual void run() = 0;
};

#endif
