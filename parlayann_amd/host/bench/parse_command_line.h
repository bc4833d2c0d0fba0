// parse_command_line.h -- the option lookups the drivers use (algorithms/bench/parse_command_line.h): a flag is present
// or not (getOption), `-name value` pairs are read by name (getOptionValue / getOptionIntValue / getOptionDoubleValue).
#pragma once
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

struct commandLine {
  std::vector<std::string> args;
  std::string usage;
  commandLine(int argc, char** argv, std::string usage_ = "bad arguments") : args(argv, argv + argc), usage(std::move(usage_)) {
    if (getOption("-h") || getOption("-help")) badArgument();
  }
  void badArgument() const {
    std::cout << "usage: " << (args.empty() ? "" : args[0]) << " " << usage << std::endl;
    exit(0);
  }
  bool getOption(const std::string& option) const {
    for (size_t i = 1; i < args.size(); i++) if (args[i] == option) return true;
    return false;
  }
  // the value after `option`, or NULL; the pointer stays valid for the life of this object
  char* getOptionValue(const std::string& option) {
    for (size_t i = 1; i + 1 < args.size(); i++) if (args[i] == option) return args[i + 1].data();
    return NULL;
  }
  std::string getOptionValue(const std::string& option, const std::string& def) {
    char* v = getOptionValue(option);
    return v ? std::string(v) : def;
  }
  long getOptionLongValue(const std::string& option, long def) {
    char* v = getOptionValue(option);
    if (!v) return def;
    char* end = nullptr;
    const long r = std::strtol(v, &end, 10);
    if (end == v) badArgument();
    return r;
  }
  int getOptionIntValue(const std::string& option, int def) { return (int)getOptionLongValue(option, def); }
  double getOptionDoubleValue(const std::string& option, double def) {
    char* v = getOptionValue(option);
    if (!v) return def;
    char* end = nullptr;
    const double r = std::strtod(v, &end);
    if (end == v) badArgument();
    return r;
  }
};
