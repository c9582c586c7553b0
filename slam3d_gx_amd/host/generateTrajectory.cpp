// generateTrajectory keyframe.txt final.g2o -- the reference's trajectory exporter (src/generateTrajectory.cpp:17-88):
// for every keyframe "id frame" of keyframe.txt, the time stamp of that frame from <data_source>/associate.txt (first
// token of line `frame`, 1-based like the file names) and the pose of vertex id from the g2o text file, written to
// trajectory.txt as the TUM line "timestamp tx ty tz qx qy qz qw" (:62-72).  No g2o: the graph file is read as text.
// Without an associate.txt the frame index stands in for the time stamp.
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "ParameterReader.h"
#include "PoseGraph.h"

using namespace std;

int main(int argc, char **argv)
{
    if (argc != 3) { cout << "generateTrajectory keyframe.txt final.g2o" << endl; return -1; }
    ParameterReader reader("./parameters.yaml");
    g_pParaReader = &reader;
    PoseGraph graph;
    if (!graph.load(argv[2])) { cout << "file does not exist" << endl; return -1; }
    ifstream fin(argv[1]);
    if (!fin) { cout << "file does not exist" << endl; return -1; }              // :33-37
    vector<string> stamps;                                                      // stamps[k] = time stamp of frame k + 1
    {
        ifstream asso((reader.GetPara("data_source") + "/associate.txt").c_str());
        string line;
        while (asso && getline(asso, line)) {
            istringstream is(line);
            string t;
            if (is >> t) stamps.push_back(t);
        }
    }
    ofstream fout("trajectory.txt");
    int id, frame;
    while (fin >> id >> frame) {
        const PoseVertex *pv = graph.vertex(id);
        if (!pv) continue;                                                      // :64-65
        double qx, qy, qz, qw;
        rot_to_quat(pv->T, qx, qy, qz, qw);
        if (frame >= 1 && (size_t)frame <= stamps.size()) fout << stamps[frame - 1] << " ";
        else fout << frame << " ";
        fout << pv->T[3] << " " << pv->T[7] << " " << pv->T[11] << " " << qx << " " << qy << " " << qz << " " << qw << " " << endl;   // :68-71
    }
    cout << "trajectory saved." << endl;
    g_pParaReader = nullptr;
    return 0;
}
