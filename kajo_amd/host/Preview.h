// Preview.h -- headless stand-in for the reference's SDL preview window, with the reference's own
// declarations (renderer/Preview.h:15-24): a concrete class, NON-virtual processEvents() / update(),
// a private constructor behind `static create(Image*, bool useOpenGL)`. hip::Scheduler is compiled
// against this header here and against the reference's renderer/Preview.h in a Kajo checkout
// (`make -C kajo_amd/host refcheck` proves the second), so nothing in HipScheduler.{h,cpp} may use
// anything beyond those two calls.
//
// The reference's window is the only stopping condition its schedulers have
// (renderer/cpu/Scheduler.cpp:74); this one "closes" after a pass budget -- setPassBudget() and the
// read-back accessors are additions of the headless driver (kajo_amd/host/Main.cpp), not part of the
// reference's interface, and the backend does not call them.
#ifndef KAJO_HOST_PREVIEW_H
#define KAJO_HOST_PREVIEW_H

#include <chrono>
#include <memory>
#include <thread>
#include <vector>

class Image;

class Preview
{
public:
    ~Preview();
    static std::unique_ptr<Preview> create(Image* image, bool useOpenGL = false);

    bool processEvents();
    void update(std::thread::id threadId, int pass, int samples, int xOffset, int yOffset, int width, int height);

    // ---- headless driver only --------------------------------------------------------------
    void setPassBudget(int passes, bool verbose = false);
    // The window "closes" at the k-th call of processEvents() (k >= 1: that call and every later one answer false) -- Esc pressed
    // mid-run (renderer/Preview.cpp:216-234) -- whatever the pass count is then. 0 = never.
    void closeAtEvent(int k) { m_closeAt = k; }
    struct Update
    {
        int pass;
        bool onCreatingThread; // update() must be called on the thread that owns the window (SDL: the main thread)
    };
    const std::vector<Update>& updates() const { return m_updates; }
    int eventCalls() const { return m_events; }
    int pass() const { return m_pass; }
    // same accounting as the reference's Preview::update (renderer/Preview.cpp:81-82): samples * width * height per call
    long long nominalSamples() const { return m_samples; }

private:
    Preview(Image* image);

    Image* m_image;
    int m_budget;
    bool m_verbose;
    int m_pass;
    long long m_samples;
    std::chrono::steady_clock::time_point m_startTime;
    int m_closeAt = 0, m_events = 0;
    std::thread::id m_owner;
    std::vector<Update> m_updates;
};

#endif
