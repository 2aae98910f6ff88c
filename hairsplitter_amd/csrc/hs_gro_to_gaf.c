/* The .gro consumer of the reference's stage 5 as a stand-alone tool: parse_split_file + merge_intervals + output_GAF
 * (create_new_contigs.cpp:1582-1590), without the external polishing tools the rest of that stage shells out to. */
#include "../../include/hairsplitter_hip.h"
int main(int argc, char** argv) { return hs_gro_to_gaf_main(argc, argv); }
