// helpers.h -- drop-in for the reference's include/stock_market_monte_carlo/helpers.h:
// the trajectory CSV writers (src/helpers.cpp:10-39) whose files python/plot_returns.py
// reads.  Host-only, identical text format; no fmt dependency.
#ifndef SMMC_DROPIN_HELPERS_H
#define SMMC_DROPIN_HELPERS_H

#include <string>
#include <vector>

void print_vector(std::vector<float> &v);                                  // src/helpers.cpp:10-16
void write_vector_file(std::string fname, std::vector<float> &v);          // :18-21: "v0,v1,...,"
// :23-39: ./outputs/<fname> with the lines "Returns,,r0,r1,...," and "Values,v0,v1,...,"
void write_data_file(std::string fname, std::vector<float> &returns, std::vector<float> &values);

#endif
