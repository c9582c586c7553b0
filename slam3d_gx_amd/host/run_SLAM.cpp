// run_SLAM [loops] -- the reference's CLI contract (src/run_SLAM.cpp:11-44): one positional argument (loop
// count), ./parameters.yaml in the working directory, results under ./data/.  Front end = GraphicEndICP, the
// MI355X plane-ICP subclass-equivalent of GraphicEnd (selected here exactly like src/run_SLAM_imageonly.cpp:21
// selects GraphicEnd2).
#include <cstdlib>
#include <iostream>

#include "GraphicEndICP.h"

using namespace std;

int main(int argc, char **argv)
{
    int nloops = 10;
    if (argc >= 2) nloops = atoi(argv[1]);
    GraphicEndICP *pGraphicEnd = new GraphicEndICP();
    pGraphicEnd->init("./parameters.yaml");
    cout << "Total loops: " << nloops << endl;
    for (int i = 0; i < nloops; i++) {
        cout << "Loop " << i << endl;
        pGraphicEnd->run();
    }
    pGraphicEnd->saveFinalResult("./data/keyframe.txt");
    cout << "keyframes: " << pGraphicEnd->keyframes().size() << endl;
    delete pGraphicEnd;
    return 0;
}
