// goss -- the `goss build-kmer-set` / `goss build-graph` executable (goss.cc:15-32).
#include "GossHost.hpp"

int main(int argc, char* argv[]) { return gosshost::gossMain(argc, argv); }
