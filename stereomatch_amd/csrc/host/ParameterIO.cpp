// ParameterIO.cpp -- see ParameterIO.h.
#include "ParameterIO.h"
#include "Error.h"
#include <cstdlib>

void CParameterIO::Add(const char *name, Kind kind, void *loc)
{
    m_index.emplace(name, m_params.size());
    m_params.push_back(Param{name, kind, loc});
}

void CParameterIO::PushParamPair(const char *name, std::string *text) { Add(name, Kind::Text, text); }
void CParameterIO::PushParamPair(const char *name, float *real) { Add(name, Kind::Real, real); }
void CParameterIO::PushParamPair(const char *name, int *integer, bool hex)
{
    Add(name, hex ? Kind::Hex : Kind::Integer, integer);
}

// A value that does not parse leaves the variable as it was (what scanf does with a failed conversion).
void CParameterIO::Param::Parse(const std::string &value) const
{
    const char *s = value.c_str();
    char *end = nullptr;
    switch (kind) {
    case Kind::Text:
        *static_cast<std::string *>(loc) = value;
        break;
    case Kind::Real: {
        float v = strtof(s, &end);
        if (end != s)
            *static_cast<float *>(loc) = v;
        break;
    }
    case Kind::Integer: {
        long v = strtol(s, &end, 10);
        if (end != s)
            *static_cast<int *>(loc) = (int)v;
        break;
    }
    case Kind::Hex:                                    // written as 0x%08x, read back the same way
        if (value.size() > 2 && s[0] == '0' && s[1] == 'x') {
            const std::string digits = value.substr(2, 8);
            unsigned long v = strtoul(digits.c_str(), &end, 16);
            if (end != digits.c_str())                 // no hex digit after "0x": the variable keeps its value, as with scanf
                *static_cast<int *>(loc) = (int)(unsigned)v;
        }
        break;
    }
}

bool CParameterIO::Param::Print(std::string &out) const
{
    char buf[64];
    switch (kind) {
    case Kind::Text:
        out = *static_cast<const std::string *>(loc);
        return !out.empty();
    case Kind::Real:
        snprintf(buf, sizeof(buf), "%f", *static_cast<const float *>(loc));
        break;
    case Kind::Integer:
        snprintf(buf, sizeof(buf), "%d", *static_cast<const int *>(loc));
        break;
    case Kind::Hex:
        snprintf(buf, sizeof(buf), "0x%08x", (unsigned)*static_cast<const int *>(loc));
        break;
    }
    out = buf;
    return true;
}

void CParameterIO::Assign(const std::string &name, const std::string &value, bool warn_bad_name)
{
    auto it = m_index.find(name);
    if (it != m_index.end()) {
        m_params[it->second].Parse(value);
        return;
    }
    // `script` and `cd` are commands of the interpreter that travel through the same pair lists
    if (warn_bad_name && name != "script" && name != "cd")
        fprintf(stderr, "Warning: parameter '%s' is not known and was ignored\n", name.c_str());
}

void CParameterIO::ReadFromStream(FILE *stream)
{
    // the whole stream is a white-space separated list name value name value ...
    std::vector<std::string> words;
    std::string word;
    for (int c; (c = getc(stream)) != EOF;) {
        bool space = c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v';
        if (!space)
            word.push_back((char)c);
        else if (!word.empty()) {
            words.push_back(word);
            word.clear();
        }
    }
    if (!word.empty())
        words.push_back(word);
    for (size_t i = 0; i + 1 < words.size(); i += 2)
        Assign(words[i], words[i + 1], true);
}

void CParameterIO::ReadFromFile(const char *filename)
{
    FILE *stream = fopen(filename, "r");
    if (!stream)
        throw CError("parameter file: could not open %s for reading", filename);
    ReadFromStream(stream);
    fclose(stream);
}

void CParameterIO::ReadFromCommandLine(int argc, const char *argv[], bool warn_bad_name)
{
    for (int i = 0; i + 1 < argc; i += 2)
        Assign(argv[i], argv[i + 1], warn_bad_name);
}

void CParameterIO::WriteToStream(FILE *stream, bool single_line)
{
    const char *separator = single_line ? " " : "\n";
    std::string text;
    for (const Param &p : m_params)
        if (p.Print(text))
            fprintf(stream, "%s %s%s", p.name.c_str(), text.c_str(), separator);
    if (single_line)
        fputc('\n', stream);
}

void CParameterIO::WriteToFile(const char *filename)
{
    const bool append = filename[0] == '+';
    const char *path = filename + (append ? 1 : 0);
    FILE *stream = fopen(path, append ? "a" : "w");
    if (!stream)
        throw CError("parameter file: could not open %s for writing", path);
    WriteToStream(stream);
    if (append)
        fputc('\n', stream);                           // blank line between appended records
    fclose(stream);
}
