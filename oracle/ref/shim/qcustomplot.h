// TEST INFRASTRUCTURE ONLY (oracle/_ref build).
// Empty stand-in so JAERO/gui_classes/qspectrumdisplay.h (included by JAERO/oqpskdemodulator.cpp:2 only for
// SPECTRUM_FFT_POWER) parses without the QCustomPlot GUI library.
#ifndef ORACLE_SHIM_QCUSTOMPLOT_H
#define ORACLE_SHIM_QCUSTOMPLOT_H
#include <QObject>
#include <QWidget>
#include <QElapsedTimer>
#include <QVector>
class QCPBars;
class QCustomPlot : public QWidget { public: explicit QCustomPlot(QWidget *parent = 0) : QWidget(parent) {} };
#endif
