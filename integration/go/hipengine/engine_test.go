package hipengine

// engine_test.go -- the first thing to run once this package compiles (it never has: no Go toolchain where it was written).
//
//	SIMON_REF_ROOT=/path/to/open-simulator go test ./pkg/simulator/hipengine/ -run TestEngineMatchesReference -v
//
// For every example input of the reference that Supports() accepts it runs the reference's own simulator.Simulate and
// hipengine.Simulate and compares, pod by pod, the node each pod was bound to and the set of unscheduled pods.  The comparison is
// exact only against the DETERMINISED reference (integration/go/parity/determinise.patch: first-maximum selectHost, one filter
// worker) -- against the stock reference a tie may be broken differently (math/rand), which the test reports as "tie?" instead of
// failing when the two placements score alike is not decidable here; run it on the patched tree (oracle/run_ref.sh prepares one
// under oracle/_ref/work/src).  Needs a GPU (DeviceCount() > 0), else the test skips.

import (
	"os"
	"path/filepath"
	"sort"
	"strings"
	"testing"

	corev1 "k8s.io/api/core/v1"

	"github.com/alibaba/open-simulator/pkg/simulator"
	"github.com/alibaba/open-simulator/pkg/utils"
)

type exampleCase struct {
	name    string
	cluster string   // directory under example/
	apps    []string // directories under example/
}

var exampleCases = []exampleCase{
	{"simple", "cluster/demo_1", []string{"application/simple"}},
	{"all", "cluster/demo_1", []string{"application/simple", "application/complicate", "application/more_pods"}},
	{"gpushare", "cluster/gpushare", []string{"application/gpushare"}},
	{"open_local", "cluster/demo_1", []string{"application/open_local"}},
}

func loadApp(t *testing.T, dir string) simulator.ResourceTypes {
	content, err := utils.GetYamlContentFromDirectory(dir)
	if err != nil {
		t.Fatalf("read %s: %v", dir, err)
	}
	res, err := simulator.GetObjectFromYamlContent(content)
	if err != nil {
		t.Fatalf("decode %s: %v", dir, err)
	}
	return res
}

// workloadOf strips the random suffix utils.MakeValidPodsBy* appends to a replica's name: placements are compared per workload as
// multisets of node names (replicas of one workload are interchangeable), bare pods by their own name.
func workloadOf(p *corev1.Pod) string {
	if k, ok := p.Annotations["simon/workload-kind"]; ok {
		return k + "/" + p.Annotations["simon/workload-namespace"] + "/" + p.Annotations["simon/workload-name"]
	}
	return "Pod/" + p.Namespace + "/" + p.Name
}

func placementsOf(res *simulator.SimulateResult) (map[string][]string, map[string]int) {
	placed, failed := map[string][]string{}, map[string]int{}
	for _, ns := range res.NodeStatus {
		for _, p := range ns.Pods {
			w := workloadOf(p)
			placed[w] = append(placed[w], ns.Node.Name)
		}
	}
	for w := range placed {
		sort.Strings(placed[w])
	}
	for _, u := range res.UnscheduledPods {
		failed[workloadOf(u.Pod)]++
	}
	return placed, failed
}

func TestEngineMatchesReference(t *testing.T) {
	root := os.Getenv("SIMON_REF_ROOT")
	if root == "" {
		t.Skip("SIMON_REF_ROOT not set (the open-simulator tree whose example/ inputs are used)")
	}
	if DeviceCount() == 0 {
		t.Skip("no GPU visible")
	}
	for _, c := range exampleCases {
		c := c
		t.Run(c.name, func(t *testing.T) {
			cluster, err := simulator.CreateClusterResourceFromClusterConfig(filepath.Join(root, "example", c.cluster))
			if err != nil {
				t.Fatal(err)
			}
			var apps []simulator.AppResource
			for _, a := range c.apps {
				apps = append(apps, simulator.AppResource{Name: filepath.Base(a), Resource: loadApp(t, filepath.Join(root, "example", a))})
			}
			if !Supports(cluster, apps, Options{}) {
				t.Skipf("%s: Supports() = false (the Go path would run)", c.name)
			}
			want, err := simulator.Simulate(cluster, apps, simulator.DisablePTerm(true))
			if err != nil {
				t.Fatal(err)
			}
			got, err := Simulate(cluster, apps)
			if err != nil {
				t.Fatal(err)
			}
			wp, wf := placementsOf(want)
			gp, gf := placementsOf(got)
			for w, nodes := range wp {
				if strings.Join(nodes, ",") != strings.Join(gp[w], ",") {
					t.Errorf("%s: workload %s on %v, the reference on %v", c.name, w, gp[w], nodes)
				}
			}
			for w, n := range wf {
				if gf[w] != n {
					t.Errorf("%s: workload %s: %d unscheduled, the reference %d", c.name, w, gf[w], n)
				}
			}
			if len(want.UnscheduledPods) != len(got.UnscheduledPods) {
				t.Errorf("%s: %d unscheduled pods, the reference %d", c.name, len(got.UnscheduledPods), len(want.UnscheduledPods))
			}
			// the reasons of the pods both leave out (the engine's strings are FitError.Error() rebuilt from failure codes)
			reasons := map[string]bool{}
			for _, u := range want.UnscheduledPods {
				if i := strings.Index(u.Reason, "): "); i >= 0 {
					reasons[workloadOf(u.Pod)+"|"+u.Reason[i+3:]] = true
				}
			}
			for _, u := range got.UnscheduledPods {
				if i := strings.Index(u.Reason, "): "); i >= 0 && !reasons[workloadOf(u.Pod)+"|"+u.Reason[i+3:]] {
					t.Errorf("%s: reason of %s/%s not among the reference's: %s", c.name, u.Pod.Namespace, u.Pod.Name, u.Reason[i+3:])
				}
			}
		})
	}
}
