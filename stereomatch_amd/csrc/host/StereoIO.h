// StereoIO.h -- command-line / script interpreter around CStereoMatcher, with the grammar of
// the reference's CStereoIO (StereoIO.cpp:168-473): commands `script <file>`, `cd <dir>`,
// `reset`, `exit`; every other line is a list of name/value pairs; `input_params <file>` loads
// more pairs, `input_data <file>` lists the frames, and a line carrying `depth_map <out.pgm>`
// runs the matcher and writes <out>.pgm and <out>.txt.
#pragma once
#include "StereoMatcher.h"
#include <string>

struct CStereoFrameDescriptor {
    std::string input_file, depth_file, truth_file, resampled_file;
    void PIOInitialize(CParameterIO &o);
};

struct CStereoIOParameters {
    // "use once" parameters, cleared on every command line (StereoIO.cpp:175)
    std::string input_params, output_params, input_data, output_data, depth_map, cost_map;
    void PIOInitialize(CParameterIO &o);
};

class CStereoIO : public CStereoMatcher, public CStereoIOParameters {
public:
    CStereoIO();
    int InterpretCommandLine(int argc, const char *argv[]);   // returns 1 on `exit`

protected:
    // what a command of the interpreter does to the rest of its line
    enum class LineFlow { GoOn, LineDone, StopScript };
    void BeginLine(int argc, const char *argv[]);
    LineFlow CommandScript(const std::string &file);
    LineFlow CommandChangeDir(const std::string &dir);
    LineFlow CommandReset(const std::string &);
    LineFlow CommandExit(const std::string &);
    void LoadLineInputs(int argc, const char *argv[]);
    bool PlanOutputs();
    void WriteLineOutputs();
    void InterpretScriptFile(const char *script_file);
    void ReadParameters();
    void WriteParameters();
    void ReadData();
    void WriteData();
    void RunMatcher();

    CParameterIO m_pio, m_dio;
    CStereoFrameDescriptor m_fd0;
    std::vector<CStereoFrameDescriptor> m_fd;
};
