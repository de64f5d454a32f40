// Preview.h -- the two Preview calls a backend makes (renderer/Preview.h:23-24 in the reference:
// bool processEvents(); void update(threadId, pass, samples, xOffset, yOffset, width, height)),
// as an interface. The reference's Preview is an SDL window; this repo's driver is headless and
// supplies PassBudgetPreview, which "closes the window" after a number of passes -- the reference
// itself has no stopping condition other than the window (renderer/cpu/Scheduler.cpp:74).
#ifndef KAJO_HOST_PREVIEW_H
#define KAJO_HOST_PREVIEW_H

#include <chrono>
#include <thread>

class Preview
{
public:
    virtual ~Preview() {}
    virtual bool processEvents() = 0;
    virtual void update(std::thread::id threadId, int pass, int samples, int xOffset, int yOffset, int width,
                        int height) = 0;
};

class PassBudgetPreview : public Preview
{
public:
    explicit PassBudgetPreview(int passes, bool verbose = false): m_budget(passes), m_verbose(verbose) {}
    bool processEvents() override { return m_pass < m_budget; }
    void update(std::thread::id, int pass, int samples, int, int, int width, int height) override;
    int pass() const { return m_pass; }
    // same accounting as Preview::update (renderer/Preview.cpp:81-82): samples * width * height per call
    long long nominalSamples() const { return m_samples; }

private:
    int m_budget;
    bool m_verbose;
    int m_pass = 0;
    long long m_samples = 0;
    std::chrono::steady_clock::time_point m_start = std::chrono::steady_clock::now();
};

#endif
