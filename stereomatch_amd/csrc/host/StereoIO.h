// StereoIO.h -- command-line / script interpreter around CStereoMatcher, with the grammar of
// the reference's CStereoIO (StereoIO.cpp:168-473): commands `script <file>`, `cd <dir>`,
// `reset`, `exit`; every other line is a list of name/value pairs; `input_params <file>` loads
// more pairs, `input_data <file>` lists the frames, and a line carrying `depth_map <out.pgm>`
// runs the matcher and writes <out>.pgm and <out>.txt.
#pragma once
#include "StereoMatcher.h"

struct CStereoFrameDescriptor {
    char input_file[1024], depth_file[1024], truth_file[1024], resampled_file[1024];
    CStereoFrameDescriptor();
    void PIOInitialize(CParameterIO &o);
};

struct CStereoIOParameters {
    // "use once" parameters, cleared on every command line (StereoIO.cpp:175)
    char input_params[1024], output_params[1024], input_data[1024], output_data[1024];
    char depth_map[1024], cost_map[1024];
    CStereoIOParameters();
    void PIOInitialize(CParameterIO &o);
};

class CStereoIO : public CStereoMatcher, public CStereoIOParameters {
public:
    CStereoIO();
    int InterpretCommandLine(int argc, const char *argv[]);   // returns 1 on `exit`

protected:
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
