// Preview.cpp -- the headless preview of this repo's driver: no window, no SDL; processEvents()
// answers "the window is still open" until the pass budget is used up (the reference's own preview
// answers it until Esc, renderer/Preview.cpp:216-234), update() keeps the reference's samples/s
// accounting (renderer/Preview.cpp:79-98: samples * width * height per call).
#include "Preview.h"

#include <cstdio>

Preview::Preview(Image* image):
    m_image(image), m_budget(0), m_verbose(false), m_pass(0), m_samples(0), m_startTime(std::chrono::steady_clock::now()),
    m_owner(std::this_thread::get_id())
{
}

Preview::~Preview() {}

std::unique_ptr<Preview> Preview::create(Image* image, bool)
{
    return std::unique_ptr<Preview>(new Preview(image));
}

void Preview::setPassBudget(int passes, bool verbose)
{
    m_budget = passes;
    m_verbose = verbose;
}

bool Preview::processEvents()
{
    m_events++;
    if (m_closeAt > 0 && m_events >= m_closeAt)
        return false;
    return m_budget <= 0 || m_pass < m_budget;
}

void Preview::update(std::thread::id, int pass, int samples, int, int, int width, int height)
{
    m_pass = pass;
    m_updates.push_back(Update{pass, std::this_thread::get_id() == m_owner});
    m_samples += (long long)samples * width * height;
    if (m_verbose) {
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - m_startTime).count();
        std::fprintf(stderr, "pass %d  %.2f s  %.1f M nominal samples/s\n", pass, s, m_samples / s / 1e6);
    }
}
