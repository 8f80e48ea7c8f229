// StereoIO.cpp -- see StereoIO.h.
#include "StereoIO.h"
#include "Error.h"
#include "ImageIO.h"
#include <cstring>
#include <string>
#include <unistd.h>

void CStereoFrameDescriptor::PIOInitialize(CParameterIO &o)
{
    o.PPPS(input_file); o.PPPS(depth_file); o.PPPS(truth_file); o.PPPS(resampled_file);
}

void CStereoIOParameters::PIOInitialize(CParameterIO &o)
{
    o.PPPS(input_params); o.PPPS(output_params); o.PPPS(input_data); o.PPPS(output_data);
    o.PPPS(depth_map); o.PPPS(cost_map);
}

CStereoIO::CStereoIO()
{
    CStereoParameters::PIOInitialize(m_pio);
    CStereoIOParameters::PIOInitialize(m_pio);
    m_fd0.PIOInitialize(m_dio);
}

namespace {

// splits a line into white-space separated tokens; '#' starts a comment (StereoIO.cpp:302-327)
std::vector<std::string> Tokenize(const char *line)
{
    std::vector<std::string> out;
    const char *p = line;
    while (*p) {
        while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
        if (!*p || *p == '#' || out.size() >= 255)
            break;
        const char *q = p;
        while (*q && *q != ' ' && *q != '\t' && *q != '\r' && *q != '\n') q++;
        out.emplace_back(p, q);
        p = q;
    }
    return out;
}

void ReadIfThere(CByteImage &img, const std::string &path, bool ok_if_not_there, int verbose)
{
    if (path.empty())
        return;
    const char *filename = path.c_str();
    if (verbose >= eVerboseFileIO)
        fprintf(stderr, "reading image %s\n", filename);
    try {
        ReadImage(img, filename);
    } catch (CError &) {
        if (!ok_if_not_there)
            throw;
        return;
    }
    int nb = img.Shape().nBands;
    if (!(nb == 1 || nb == 4))                      // StereoIO.cpp:76-81
        throw CError("ReadImage(%s): number of bands (%d) not supported by stereo matcher", filename, nb);
}

}  // namespace

// One line of the grammar (StereoIO.cpp:168-279 defines what a line means; StereoIO.h here lists it).
// A line is handled in phases, each a method of its own:
//   BeginLine      the use-once file names are cleared and the line's pairs are read a first time
//   a command      `script`, `cd`, `reset`, `exit` (table below); only `reset` lets the line go on
//   LoadLineInputs input_params file, then the line's pairs again (the line overrides the file), frames
//   PlanOutputs    where the depth map and the result file go; false = the line only set parameters
//   RunMatcher, WriteLineOutputs
int CStereoIO::InterpretCommandLine(int argc, const char *argv[])
{
    if (argc < 2 || !argv[1][0])
        throw CError("No command is given on command line.");
    BeginLine(argc, argv);

    struct Command { const char *word; LineFlow (CStereoIO::*run)(const std::string &argument); };
    static const Command commands[] = {
        {"script", &CStereoIO::CommandScript}, {"cd", &CStereoIO::CommandChangeDir},
        {"reset", &CStereoIO::CommandReset}, {"exit", &CStereoIO::CommandExit}};
    const std::string word = argv[1], argument = argc > 2 ? argv[2] : "";
    for (const Command &c : commands)
        if (word == c.word) {
            LineFlow flow = (this->*c.run)(argument);
            if (flow != LineFlow::GoOn)
                return flow == LineFlow::StopScript ? 1 : 0;
        }

    LoadLineInputs(argc, argv);
    if (!PlanOutputs())
        return 0;
    RunMatcher();
    WriteLineOutputs();
    return 0;
}

void CStereoIO::BeginLine(int argc, const char *argv[])
{
    static_cast<CStereoIOParameters &>(*this) = CStereoIOParameters();
    m_pio.ReadFromCommandLine(argc - 1, argv + 1);
}

CStereoIO::LineFlow CStereoIO::CommandScript(const std::string &file)
{
    InterpretScriptFile(file.c_str());
    return LineFlow::LineDone;
}

CStereoIO::LineFlow CStereoIO::CommandChangeDir(const std::string &dir)
{
    if (chdir(dir.c_str()) != 0)
        throw CError("Could not cd to %s", dir.c_str());
    return LineFlow::LineDone;
}

CStereoIO::LineFlow CStereoIO::CommandReset(const std::string &)
{
    ReInitializeAlgParams();
    return LineFlow::GoOn;
}

CStereoIO::LineFlow CStereoIO::CommandExit(const std::string &) { return LineFlow::StopScript; }

void CStereoIO::LoadLineInputs(int argc, const char *argv[])
{
    ReadParameters();
    m_pio.ReadFromCommandLine(argc - 1, argv + 1, false);
    ResetOutputParams();
    ReadData();
}

bool CStereoIO::PlanOutputs()
{
    const int n_frames = (int)m_fd.size();
    if (!depth_map.empty() && frame_ref >= 0 && frame_ref < n_frames)
        m_fd[frame_ref].depth_file = depth_map;
    if (depth_map.empty() && output_params.empty())
        return false;
    if (frame_ref >= n_frames || frame_match >= n_frames)
        throw CError("Data has not yet been read in");
    if (evaluate_only)                                  // the map to evaluate comes from the file
        ReadIfThere(m_frame[frame_ref].depth_image, depth_map, false, verbose);
    if (output_params.empty()) {                        // results/x.pgm -> results/x.txt
        size_t stem = depth_map.size() >= 3 ? depth_map.size() - 3 : 0;
        output_params = depth_map.substr(0, stem) + "txt";
    }
    return true;
}

void CStereoIO::WriteLineOutputs()
{
    WriteData();
    WriteParameters();
    if (verbose >= eVerboseSummary)
        fprintf(stderr, "  writing %s\n", output_params.c_str());
}

void CStereoIO::RunMatcher()
{
    if (!depth_map.empty()) {
        // cost_map ("cost map (DSI), for debugging", StereoIO.h:85 -- declared but unused in the
        // reference): here it names the file the aggregated DSI is written to, one PGM per level
        // ("..._%03d.pgm") or the whole volume as one .pmf; the DSI has to exist for that
        bool fused = allow_fused;
        if (!cost_map.empty())
            allow_fused = false;
        ComputeCorrespondence();
        allow_fused = fused;
        if (!cost_map.empty() && !evaluate_only)
            WriteCosts(cost_map.c_str());
    }
    if (!output_params.empty())
        Evaluate();
}

void CStereoIO::InterpretScriptFile(const char *script_file)
{
    FILE *stream = fopen(script_file, "r");
    if (!stream)
        throw CError("InterpretScriptFile: could not open %s", script_file);
    if (verbose >= eVerboseScriptFile)
        fprintf(stderr, "Interpreting script file %s\n", script_file);
    char line[4096];
    try {
        while (fgets(line, sizeof(line), stream)) {
            if (verbose >= eVerboseScriptFile)
                fputs(line, stderr);
            std::vector<std::string> tok = Tokenize(line);
            if (tok.empty())
                continue;
            std::vector<const char *> argv(1, nullptr);
            for (const std::string &t : tok) argv.push_back(t.c_str());
            if (InterpretCommandLine((int)argv.size(), argv.data()))
                break;
        }
    } catch (...) {
        fclose(stream);
        throw;
    }
    fclose(stream);
}

void CStereoIO::ReadParameters()
{
    if (!input_params.empty())
        m_pio.ReadFromFile(input_params.c_str());
}

void CStereoIO::WriteParameters()
{
    if (!output_params.empty())
        m_pio.WriteToFile(output_params.c_str());
}

void CStereoIO::ReadData()
{
    if (input_data.empty())
        return;
    FILE *stream = fopen(input_data.c_str(), "r");
    if (!stream)
        throw CError("ReadData: could not open %s", input_data.c_str());
    char line[4096];
    m_fd.clear();
    m_frame.clear();
    try {
        while (fgets(line, sizeof(line), stream)) {
            std::vector<std::string> tok = Tokenize(line);
            if (tok.empty())
                continue;
            std::vector<const char *> argv;
            for (const std::string &t : tok) argv.push_back(t.c_str());
            m_fd0 = CStereoFrameDescriptor();
            m_dio.ReadFromCommandLine((int)argv.size(), argv.data());
            if ((int)m_fd.size() == frame_ref && !depth_map.empty())
                m_fd0.depth_file = depth_map;
            m_fd.push_back(m_fd0);
            CStereoFrame fr;
            ReadIfThere(fr.input_image, m_fd0.input_file, false, verbose);
            ReadIfThere(fr.depth_image, m_fd0.depth_file, true, verbose);
            ReadIfThere(fr.truth_image, m_fd0.truth_file, false, verbose);
            m_frame.push_back(fr);
        }
    } catch (...) {
        fclose(stream);
        throw;
    }
    fclose(stream);
}

void CStereoIO::WriteData()
{
    FILE *stream = nullptr;
    if (!output_data.empty()) {
        stream = fopen(output_data.c_str(), "w");
        if (!stream)
            throw CError("WriteData: could not open %s", output_data.c_str());
    }
    for (size_t i = 0; i < m_fd.size(); i++) {
        m_fd0 = m_fd[i];
        CStereoFrame &fr = m_frame[i];
        if (!evaluate_only && !m_fd0.depth_file.empty() && fr.depth_image.Shape().height > 0) {
            if (verbose >= eVerboseFileIO)
                fprintf(stderr, "writing image %s\n", m_fd0.depth_file.c_str());
            WriteImage(fr.depth_image, m_fd0.depth_file.c_str());
        }
        if (!m_fd0.resampled_file.empty() && fr.resampled_image.Shape().height > 0) {   // StereoIO.cpp:95-112
            if (verbose >= eVerboseFileIO)
                fprintf(stderr, "writing image %s\n", m_fd0.resampled_file.c_str());
            WriteImage(fr.resampled_image, m_fd0.resampled_file.c_str());
        }
        if (stream)
            m_dio.WriteToStream(stream, true);
    }
    if (stream)
        fclose(stream);
}
