// StereoIO.cpp -- see StereoIO.h.
#include "StereoIO.h"
#include "Error.h"
#include "ImageIO.h"
#include <cstring>
#include <string>
#include <unistd.h>

CStereoFrameDescriptor::CStereoFrameDescriptor() { memset(this, 0, sizeof(*this)); }

void CStereoFrameDescriptor::PIOInitialize(CParameterIO &o)
{
    o.PPPS(input_file); o.PPPS(depth_file); o.PPPS(truth_file); o.PPPS(resampled_file);
}

CStereoIOParameters::CStereoIOParameters() { memset(this, 0, sizeof(*this)); }

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

void ReadIfThere(CByteImage &img, const char *filename, bool ok_if_not_there, int verbose)
{
    if (!filename[0])
        return;
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

int CStereoIO::InterpretCommandLine(int argc, const char *argv[])
{
    const char *command = argc > 1 ? argv[1] : nullptr;
    const char *argument = argc > 2 ? argv[2] : "";
    static_cast<CStereoIOParameters &>(*this) = CStereoIOParameters();
    m_pio.ReadFromCommandLine(argc - 1, argv + 1);
    if (!command || !command[0])
        throw CError("No command is given on command line.");

    if (strcmp("script", command) == 0) {
        InterpretScriptFile(argument);
        return 0;
    }
    if (strcmp("cd", command) == 0) {
        if (chdir(argument))
            throw CError("Could not cd to %s", argument);
        return 0;
    }
    if (strcmp("reset", command) == 0)
        ReInitializeAlgParams();
    if (strcmp("exit", command) == 0)
        return 1;

    ReadParameters();
    m_pio.ReadFromCommandLine(argc - 1, argv + 1, false);   // the line overrides the file
    ResetOutputParams();
    ReadData();
    if (frame_ref >= 0 && frame_ref < (int)m_fd.size() && depth_map[0])
        strcpy(m_fd[frame_ref].depth_file, depth_map);
    if (!depth_map[0] && !output_params[0])
        return 0;
    if (frame_ref >= (int)m_fd.size() || frame_match >= (int)m_fd.size())
        throw CError("Data has not yet been read in");
    if (evaluate_only)                                        // StereoIO.cpp:254-257: the map to evaluate
        ReadIfThere(m_frame[frame_ref].depth_image, depth_map, false, verbose);
    if (!output_params[0]) {                                  // results/x.pgm -> results/x.txt
        strcpy(output_params, depth_map);
        strcpy(&output_params[strlen(output_params) - 3], "txt");
    }
    RunMatcher();
    WriteData();
    WriteParameters();
    if (verbose >= eVerboseSummary)
        fprintf(stderr, "  writing %s\n", output_params[0] ? output_params : depth_map);
    return 0;
}

void CStereoIO::RunMatcher()
{
    if (depth_map[0]) {
        // cost_map ("cost map (DSI), for debugging", StereoIO.h:85 -- declared but unused in the
        // reference): here it names the file the aggregated DSI is written to, one PGM per level
        // ("..._%03d.pgm") or the whole volume as one .pmf; the DSI has to exist for that
        bool fused = allow_fused;
        if (cost_map[0])
            allow_fused = false;
        ComputeCorrespondence();
        allow_fused = fused;
        if (cost_map[0] && !evaluate_only)
            WriteCosts(cost_map);
    }
    if (output_params[0])
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
    if (input_params[0])
        m_pio.ReadFromFile(input_params);
}

void CStereoIO::WriteParameters()
{
    if (output_params[0])
        m_pio.WriteToFile(output_params);
}

void CStereoIO::ReadData()
{
    if (!input_data[0])
        return;
    FILE *stream = fopen(input_data, "r");
    if (!stream)
        throw CError("ReadData: could not open %s", input_data);
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
            if ((int)m_fd.size() == frame_ref && depth_map[0])
                strcpy(m_fd0.depth_file, depth_map);
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
    if (output_data[0]) {
        stream = fopen(output_data, "w");
        if (!stream)
            throw CError("WriteData: could not open %s", output_data);
    }
    for (size_t i = 0; i < m_fd.size(); i++) {
        m_fd0 = m_fd[i];
        CStereoFrame &fr = m_frame[i];
        if (!evaluate_only && m_fd0.depth_file[0] && fr.depth_image.Shape().height > 0) {
            if (verbose >= eVerboseFileIO)
                fprintf(stderr, "writing image %s\n", m_fd0.depth_file);
            WriteImage(fr.depth_image, m_fd0.depth_file);
        }
        if (m_fd0.resampled_file[0] && fr.resampled_image.Shape().height > 0) {   // StereoIO.cpp:95-112
            if (verbose >= eVerboseFileIO)
                fprintf(stderr, "writing image %s\n", m_fd0.resampled_file);
            WriteImage(fr.resampled_image, m_fd0.resampled_file);
        }
        if (stream)
            m_dio.WriteToStream(stream, true);
    }
    if (stream)
        fclose(stream);
}
