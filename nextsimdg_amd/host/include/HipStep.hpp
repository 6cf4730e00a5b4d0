// HipStep.hpp -- IModelStep implementation that runs the per-element column physics of one time
// step as ONE HIP kernel launch through the C ABI (include/nsdg.h), replacing the element loop of
// DevStep::iterate (core/src/DevStep.cpp:14-23).
//
// Ownership: the structure owns the host SoA planes; HipStep borrows it (non-owning pointer, like
// DevStep::pStructure, core/src/include/DevStep.hpp:35), owns the device copies and the nsdg context.
// Fields stay resident in HBM between steps: they are uploaded at start() and downloaded at stop() /
// writeRestartFile() / syncToHost() only.
#pragma once
#include <vector>

#include "Iterator.hpp"

struct nsdg_ctx;

namespace Nextsim {

class HipStep : public IModelStep {
public:
    HipStep();
    ~HipStep() override;
    HipStep(const HipStep&) = delete;
    HipStep& operator=(const HipStep&) = delete;

    void setInitialData(IStructure& dataStructure) override { pStructure = &dataStructure; }
    void writeRestartFile(const std::string& filePath) override;
    void init() override;
    void start(const Iterator::TimePoint& startTime) override;
    void iterate(const Iterator::Duration& dt) override;
    void stop(const Iterator::TimePoint& stopTime) override;

    //! Copies the prognostic fields (and newice) back into the structure's host planes.
    void syncToHost();
    //! Number of kernel launches issued so far (one per iterate()).
    long launches() const override { return m_launches; }

private:
    void upload();
    void release();
    IStructure* pStructure = nullptr;
    nsdg_ctx* ctx = nullptr;
    double* d_block = nullptr; // one allocation, 15 planes of n doubles
    std::size_t n = 0;
    bool resident = false;
    long m_launches = 0;
};

} // namespace Nextsim
