#include "../../espnet_amd/csrc/host_io.cpp"
#include <dirent.h>
#include <string>
int main() {
  DIR* d = opendir("/tmp/fz");
  std::vector<std::string> names;
  while (dirent* e = readdir(d)) if (e->d_name[0] != '.') names.push_back(std::string("/tmp/fz/") + e->d_name);
  closedir(d);
  int ok = 0, rej = 0, ioerr = 0;
  for (auto& nm : names) {
    const char* paths[1] = {nm.c_str()};
    EmWavInfo info;
    int rc = em_wav_probe(paths, 1, &info, 1);
    if (rc != EM_OK) { ++rej; continue; }
    std::vector<float> row((size_t)info.frames + 7);
    rc = em_wav_load_rows(paths, &info, 1, row.data(), info.frames + 7, 1);
    if (rc == EM_OK) ++ok; else ++ioerr;
  }
  printf("files %zu ok %d rejected %d decode-errors %d\n", names.size(), ok, rej, ioerr);
}
